#!/usr/bin/env python3
"""bench.py -- GCANet hot-path throughput on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: launched by torch.distributed.run, one rank per GPU, RCCL gradient all-reduce)

A "step" = forward + backward (+ gradient all-reduce + Adam update) of the hot-path module
(gcanet_amd.dgcnn.PrimitivesEmbeddingDGCNGn: 3x [kNN -> fused EdgeConv] + per-point heads +
normal-feature EdgeConv + embedding + offset module, M4:634-747) over ONE batch of synthetic clouds
already resident in HBM: BASELINE config 2 (batch 8 clouds/GPU, N=8192, k=64, bf16 MFMA + bf16
autocast for the per-point GEMMs).  Prints ONE JSON line (rank 0) with the whole-job clouds/s, a
`roofline` object for the dominant hand-written kernel (timed live with HIP events on its launch
stream) and a `cpu_baseline` object (oracle/ref_model.py, a bounded sample, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# (Round 3 note: ROCm 7.2 replays a captured graph through a "packet capture" fast path, 0.12 ms of host time per step
# instead of 5.4.  That path faults on a replay after the queue has gone idle IF the graph holds a memset node -- found
# with tools/debug/graph_trigger5.py; the library now zero-fills with a kernel (csrc/common.h: fill_dev) and the fast
# path is safe again.  DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 in the environment turns it off.)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="clouds per GPU (BASELINE config 2)")
    ap.add_argument("--points", type=int, default=8192)
    ap.add_argument("--k", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true",
                    help="launch every kernel of the timed steps from the host instead of replaying one captured HIP graph")
    ap.add_argument("--no-full", action="store_true", help="skip the second workload (literal full forward_train + losses)")
    ap.add_argument("--rehearse-cpu", action="store_true",
                    help="launcher/collective rehearsal on a box WITHOUT a GPU: gloo ranks, a tiny torch stand-in "
                         "module instead of the HIP hot path; the JSON line is marked invalid as a measurement")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="rehearsal only: allow more ranks than GPUs (ranks share devices)")
    return ap.parse_args(argv)


def launch_ranks(args):
    """`python bench.py --gpus N` outside torchrun: start N ranks (one per GPU) as CHILD processes of a parent that has
    not touched the GPU (nothing here imports torch), relay rank 0's JSON line, exit non-zero if any rank failed.
    The reference's multi-device entry is single-process nn.DataParallel (trainer_new.py:94-96)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line)
    sys.stdout.flush()
    if proc.returncode != 0 or line is None:
        print("bench.py: %d-rank launch failed (rc %d, json %s)" % (args.gpus, proc.returncode, line is not None),
              file=sys.stderr)
        sys.exit(proc.returncode or 1)
    sys.exit(0)


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ:
    _a = parse_args()
    if _a.gpus > 1:
        launch_ranks(_a)          # never returns

import torch  # noqa: E402

PEAK_TFLOPS = {"bf16_mfma": 2500.0, "f32_mfma": 157.3}   # MI355X_MICROARCH.md dense peaks
PEAK_HBM_GBS = 8000.0


def synth_clouds(cloud_ids, N, device):
    """xyz ~ U[0,1)^3 with seed 1234+cloud_id, unit normals."""
    pts, nrm = [], []
    for cid in cloud_ids:
        g = torch.Generator().manual_seed(1234 + int(cid))
        pts.append(torch.rand(N, 3, generator=g))
        nrm.append(torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1))
    return torch.stack(pts).to(device), torch.stack(nrm).to(device)


_INV_N = {}


class _SumMeanSquares(torch.autograd.Function):
    """sum_i mean(v_i^2) over the model's output tensors with multi-tensor kernels: one norm pass forward, one scaling
    pass backward (the per-tensor float()/pow/mean chain and its autograd nodes were ~45 launches per step)."""

    @staticmethod
    def forward(ctx, *vs):
        ctx.save_for_backward(*vs)
        norms = torch.stack(torch._foreach_norm(vs)).float()
        key = (tuple(v.numel() for v in vs), norms.device)
        if key not in _INV_N:
            _INV_N[key] = torch.tensor([1.0 / v.numel() for v in vs], dtype=torch.float32, device=norms.device)
        return (norms * norms * _INV_N[key]).sum()

    @staticmethod
    def backward(ctx, g):
        vs = ctx.saved_tensors
        grads = torch._foreach_mul(vs, [2.0 / v.numel() for v in vs])
        torch._foreach_mul_(grads, g)
        return tuple(grads)


def loss_of(out):
    """Synthetic objective of the benchmark: sum over the outputs of mean(v^2) (== sum(v.float().pow(2).mean()))."""
    vs = tuple(out.values())
    if vs[0].is_cuda:                  # one launch each way (csrc/heads.hip) instead of five reductions + two scaling passes
        from gcanet_amd.losses import sum_mean_squares
        return sum_mean_squares(*vs)
    return _SumMeanSquares.apply(*vs)


def kernel_model(tag):
    """Algorithmic FLOPs per launch + roofline of one of our kernels, from its timing tag."""
    import re
    kv = {k: int(v) for k, v in re.findall(r"(\w+)=(\d+)", tag)}
    if tag.startswith("knn_model"):
        # 2*B*N^2*C distance FLOPs (exact f32); f32 vector/MFMA peak is the honest ceiling.  What the entry
        # point EXECUTES since round 2 (feature space, C >= 32): a bf16 MFMA filter over all pairs (Cp + 16 deep: the
        # threshold rides in an extra k-step) + the same over a 1-in-8 sample + ~200 exact f32 keys per query
        B, N, C = kv["B"], kv["N"], kv["C"]
        m = dict(flops=2.0 * B * N ** 2 * C, bound="mfma", peak=PEAK_TFLOPS["f32_mfma"])
        if C >= 32:
            Np = (N + 127) // 128 * 128
            m["executed_flops"] = 2.0 * B * Np * Np * (C + 16) + 2.0 * B * Np * (N // 8) * C
            m["executed_peak"] = PEAK_TFLOPS["bf16_mfma"]
        return m
    if tag.startswith("edgeconv_fwd"):
        # grouped (N*k, 2C) x (2C, Cout) contraction on bf16 MFMA; executed: the centre half once per point, not per edge
        m = dict(flops=2.0 * kv["B"] * kv["N"] * kv["k"] * 2 * kv["C"] * kv["Cout"], bound="mfma", peak=PEAK_TFLOPS["bf16_mfma"])
        m["executed_flops"] = 2.0 * kv["B"] * kv["N"] * (kv["k"] + 1) * kv["C"] * kv["Cout"]
        m["executed_peak"] = PEAK_TFLOPS["bf16_mfma"]
        return m
    return None


def pmc_traffic(tag):
    """HBM bytes per call of the entry point behind `tag` (all the kernels it launches, summed), from the committed
    rocprofv3 PMC passes of this same command (profiles/r03_pmc_traffic.json: separate --pmc FETCH_SIZE / WRITE_SIZE
    runs, gfx950 read correction applied; tools/pmc_traffic.py).  Counters cannot be read from inside the process,
    hence the file; null if absent or if the profile was taken at another shape."""
    path = os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")
    if not os.path.exists(path) or "B=8" not in tag or "N=8192" not in tag:
        return None
    ks = json.load(open(path))["kernels"]
    parts = {
        "knn_model[B=8,C=64": ["gcn::knnf_colsum_kernel", "gcn::knnf_prep_kernel", "gcn::knnf_stream_kernel<4, 0>",
                               "gcn::knnf_stream_kernel<4, 1>", "gcn::knnf_keys_kernel<64>", "gcn::knnf_rank_kernel<64>",
                               "gcn::knnf_list_kernel", "gcn::knnf_fallback_kernel<64>", "gcn::knnf_fallback_merge_kernel", "gcn::knnf_transpose_kernel",
                               "gcn::knn_mfma16_kernel<64, 64, 1, true>"],
        "knn_model[B=8,C=6,": ["gcn::knnn_prep_kernel", "gcn::knnn_sample_kernel", "gcn::knnn_filter_kernel",
                               "gcn::knnn_rerank_kernel"],
        "edgeconv_fwd[B=8,N=8192,k=64,C=64,Cout=128": ["gcn::edgeconv_center_kernel<4, 4>",
                                                       "gcn::edgeconv_fwd_q_kernel<4, 4, 2, true, true, true"],
    }
    for pre, names in parts.items():
        if tag.startswith(pre):
            tot, found = 0, False
            for kn in names:
                for full, rec in ks.items():
                    if full.startswith(kn):
                        tot += rec["hbm_bytes_corrected"]
                        found = True
                        break
            return tot if found else None
    return None


def full_workload_roofline():
    """`roofline` of the literal step's dominant own kernel, from the committed profiles of tools/full_profile.py
    (profiles/r03_full_workload_kernel_stats.csv: rocprofv3 --kernel-trace --stats; r03_full_pmc_traffic.json: separate
    --pmc FETCH_SIZE / WRITE_SIZE passes).  The kernels of that step are gathers and scans over irregular lists: the
    bound named is HBM, `achieved` is the PMC traffic over the kernel's average duration.  None without the files."""
    import csv
    sp = os.path.join(ROOT, "profiles", "r03_full_workload_kernel_stats.csv")
    tp = os.path.join(ROOT, "profiles", "r03_full_pmc_traffic.json")
    if not (os.path.exists(sp) and os.path.exists(tp)):
        return None
    rows = [r for r in csv.DictReader(open(sp)) if "gcn::" in r["Name"][:20]]
    if not rows:
        return None
    top = max(rows, key=lambda r: int(r["TotalDurationNs"]))
    name = top["Name"].strip('"').replace("void ", "").split("(")[0]
    rec = json.load(open(tp))["kernels"].get(name)
    avg_ms = float(top["AverageNs"]) / 1e6
    out = {"kernel": name, "bound": "hbm", "avg_launch_ms": round(avg_ms, 4), "peak": PEAK_HBM_GBS, "unit": "GB/s",
           "source": "profiles/r03_full_workload_kernel_stats.csv + r03_full_pmc_traffic.json (not measured live)"}
    if rec:
        ach = rec["hbm_bytes_corrected"] / (avg_ms * 1e-3) / 1e9
        out.update(traffic=rec["hbm_bytes_corrected"], achieved=round(ach, 1), frac=round(ach / PEAK_HBM_GBS, 4))
    else:
        out.update(traffic=None, achieved=None, frac=None)
    if "knnf_keys" in name:
        out["note"] = ("exact f32 keys of ~200 surviving candidates per query: an L2-RESIDENT row gather (3.4 GB of 256-byte rows "
                       "per launch ~ 15 TB/s ~ 44 % of the aggregate L2 rate); HBM sees the 8 x 2 MB clouds once plus the key lists")
    return out


def cpu_baseline(N, k, seconds_budget=30.0):
    """The same hot-path step (fwd+bwd, fp32) through the CPU oracle on ONE cloud; all host cores.  One untimed
    warm-up, then best of up to 5 timed runs inside a ~30-s budget so that the default run stays short."""
    from gcanet_amd import dgcnn
    from oracle import ref_model as R
    torch.manual_seed(0)
    m = dgcnn.PrimitivesEmbeddingDGCNGn(nn_nb=k, dtype="f32")
    sd = {n_: p_.detach().clone().requires_grad_(p_.dtype.is_floating_point) for n_, p_ in m.state_dict().items()}
    pts, nrm = synth_clouds([0], N, "cpu")
    threads = torch.get_num_threads()
    knn_fn = lambda x, kk, metric: R.knn_torch(x, kk, metric)

    def one():
        t0 = time.time()
        out, _ = R.hot_path(sd, pts, nrm, k, knn_fn=knn_fn)
        loss_of(out).backward()
        return time.time() - t0

    t_all = time.time()
    one()                                                   # warm-up (allocator, thread pool, oneDNN primitives)
    best, runs = None, 0
    while runs < 5 and (runs == 0 or (time.time() - t_all) < seconds_budget):
        dt = one()
        best = dt if best is None else min(best, dt)
        runs += 1
    return {"value": round(1.0 / best, 5), "unit": "clouds/s", "cores": threads, "kind": "port",
            "sample": "1 cloud N=%d k=%d, full hot path fwd+bwd fp32 via oracle/ref_model.hot_path "
                      "(torch CPU restatement of M4:634-747), 1 warm-up + best of %d" % (N, k, runs)}


def blob_clouds(cloud_ids, N, device, blobs=64, sigma=0.004, with_centres=False):
    """Clouds made of `blobs` tight clusters (class-consistent geometry), so that forward_grouping finds proposals even
    with random-init weights: xyz = centre + sigma*N(0,1), normals = the blob's direction + noise; the blob id is the
    instance label."""
    pts, nrm, lab, ctr = [], [], [], []
    for cid in cloud_ids:
        g = torch.Generator().manual_seed(4321 + int(cid))
        centres = torch.rand(blobs, 3, generator=g) * 0.9 + 0.05
        ctr.append(centres)
        dirs = torch.nn.functional.normalize(torch.randn(blobs, 3, generator=g), dim=-1)
        which = torch.arange(N) % blobs
        pts.append(centres[which] + sigma * torch.randn(N, 3, generator=g))
        nrm.append(torch.nn.functional.normalize(dirs[which] + 0.05 * torch.randn(N, 3, generator=g), dim=-1))
        lab.append(which)
    if with_centres:
        return torch.stack(pts).to(device), torch.stack(nrm).to(device), torch.stack(lab).to(device), torch.stack(ctr).to(device)
    return torch.stack(pts).to(device), torch.stack(nrm).to(device), torch.stack(lab).to(device)


def full_workload(args, dev, steps=5, warmup=2):
    """Second workload: the LITERAL forward_train of the reference (M4:634-777) -- hot path ->
    forward_grouping on the device -> proposal cap -> clusters_voxelization -> sparse-conv instance head -- plus the
    losses that call the hot-path ops (utils/loss_utils.py:203-257,308-435) and a per-point NLL, backward and Adam, on
    blob clouds so that proposals exist.  Reported beside the headline number, never part of it."""
    from gcanet_amd.gcanet import GCANet
    from gcanet_amd.layers import CastCache
    from gcanet_amd.losses import compute_embedding_loss, instance_loss
    B, N, k = args.batch, args.points, args.k
    torch.manual_seed(0)
    net = GCANet(nn_nb=k, dtype="bf16", grouping_cfg=GROUPING_CFG).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=True)
    casts = CastCache(net.point_net, pad_k={net.point_net.conv3.weight: (net.point_net.conv3.weight.shape[1] + 15) // 16 * 16})
    pts, nrm, lab = blob_clouds(range(B), N, dev)
    blobs = int(lab.max()) + 1
    inst = (lab + torch.arange(B, device=dev).view(B, 1) * blobs).reshape(-1)          # global instance ids
    pointnum = torch.bincount(inst, minlength=B * blobs).int()
    inst_cls = (torch.arange(B * blobs, device=dev) % blobs % 9 + 1).long()            # foreground classes 1..9
    sem = inst_cls[inst]                                                               # (B*N,)
    rand = (torch.full((3,), 0.5), torch.full((3,), 0.5))
    info = {}
    from gcanet_amd.layers import ZeroArena
    own_arena = ZeroArena(dev) if ZeroArena.live is None else None     # the headline step's arena when bench.py made one

    def step():
        if ZeroArena.live is not None:
            ZeroArena.live.begin_step()
        opt.zero_grad(set_to_none=True)
        casts.refresh()
        def point_losses(out):      # the losses on per-point predictions: enqueued before forward_grouping's host wait
            with torch.autocast("cuda", enabled=False):
                tp = out["type_per_point"]
                return compute_embedding_loss(out["output_feats"].float(), lab, num_labels=blobs)[0].sum() \
                    + torch.nn.functional.nll_loss(tp.float().reshape(-1, tp.shape[-1]), sem) \
                    + out["pt_offsets"].float().abs().mean()

        with torch.autocast("cuda", dtype=torch.bfloat16):
            (type_pp, param_pp, sem_scores, off, ibi, cls_s, iou_s, mask_s, pidx, poff, feats, lpt) = \
                net(pts, nrm, rand=rand, early=point_losses)
        loss = lpt + instance_loss(cls_s.float(), mask_s.float(), iou_s.float(), pidx, poff, inst, pointnum, inst_cls, ibi)
        loss.backward()
        opt.step()
        info.update(proposals=int(poff.shape[0]) - 1, members=int(pidx.shape[0]), loss=float(loss.detach()))

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    if own_arena is not None:
        own_arena.close()
    return {"workload": "literal forward_train (M4:634-777: hot path + device forward_grouping + clusters_voxelization + "
                        "sparse instance head) + embedding/instance/NLL/offset losses + backward + Adam; %d blob clouds "
                        "N=%d k=%d (%d blobs each)" % (B, N, k, blobs),
            "ms_per_step": round(dt * 1e3, 3), "clouds_per_s": round(B / dt, 2), "steps": steps, "warmup": warmup,
            "proposals": info.get("proposals"), "members": info.get("members"), "loss": info.get("loss")}


def cfg5_workload(dev, B=4, N=16384, k=64, steps=3, warmup=2):
    """Third workload: BASELINE configs[4] on ONE GPU's share (16 clouds over 4 GPUs = 4 clouds, N=16384, k=64, C=256,
    attention in IEEE half): the hot path (M4:634-747) + a 256 -> 128 EdgeConv block on the encoder's 256-wide per-point
    features (the C=256 matrix-core kernel) + one pre-norm Transformer layer over each cloud's 16384 tokens
    (models/transformer.py:36-91, dim 256, 8 heads, fp16 flash attention forward and backward) + a two-layer QueryDecoder
    (models/query_decoder.py:104-239: 100 queries cross-attending 16384 points per cloud), forward + backward + Adam.
    The reference defines no model that joins these (its GCANet uses neither attention stack); the pieces and shapes are
    the configuration's.  Reported beside the headline number, never part of it."""
    from gcanet_amd import dgcnn
    from gcanet_amd.layers import CastCache
    from gcanet_amd.query_decoder import QueryDecoder
    from gcanet_amd.transformer import Transformer
    torch.manual_seed(0)
    net = dgcnn.PrimitivesEmbeddingDGCNGn(nn_nb=k, dtype="bf16").to(dev)
    net.keep_xf = True
    wide = dgcnn._EdgeConvLayer(512, 128, torch.nn.GroupNorm(2, 128)).to(dev)
    tr = Transformer(dim=256, depth=1, heads=8, dim_head=32, mlp_dim=512, dropout=0.0, precision="fp16").to(dev)
    qd = QueryDecoder(num_layer=2, num_query=100, num_class=10, in_channel=256, d_model=256, nhead=8, hidden_dim=512,
                      precision="fp16").to(dev)
    params = [p for m_ in (net, wide, tr, qd) for p in m_.parameters()]
    opt = torch.optim.Adam(params, lr=1e-3, fused=True)
    casts = CastCache(net, pad_k={net.conv3.weight: (net.conv3.weight.shape[1] + 15) // 16 * 16})
    pts, nrm = synth_clouds(range(B), N, dev)
    offs = [i * N for i in range(B + 1)]
    info = {}

    def step():
        from gcanet_amd.layers import ZeroArena
        if ZeroArena.live is not None:
            ZeroArena.live.begin_step()
        opt.zero_grad(set_to_none=True)
        casts.refresh()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = net(pts, nrm)
            xf = net.last_xf                                                       # (B,N,256)
            gn = wide._modules["1"]
            xw, _ = dgcnn.edge_conv_pm(xf.float(), net.encoder.last_idx[2], wide._modules["0"].weight, gn, "f16", want_cm=False)
        # the attention stacks as configs[4] names them: IEEE half -- the Linear layers under fp16 autocast (the library's
        # f32 GEMMs were a third of this step), LayerNorm / softmax statistics / losses in f32, attention cores fp16 flash
        with torch.autocast("cuda", dtype=torch.float16):
            tok = tr(xf.float())                                                   # (B,N,256)
            dec = qd(tok.reshape(B * N, 256), offs)
        tok = tok.float()
        dec = {k_: ([m_.float() for m_ in v] if isinstance(v, list) else v.float()) for k_, v in dec.items()}
        loss = loss_of(out) + xw.pow(2).mean() + tok.pow(2).mean() + dec["labels"].pow(2).mean() \
            + dec["scores"].pow(2).mean() + dec["parameters"].pow(2).mean() + sum(m_.pow(2).mean() for m_ in dec["masks"]) / B
        loss.backward()
        opt.step()
        info["loss"] = loss.detach()

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"workload": "BASELINE configs[4], one GPU's share: %d clouds N=%d k=%d: hot path + EdgeConv 256->128 (IEEE-half operands on the matrix cores) + "
                        "Transformer layer (dim 256, 8 heads) + QueryDecoder (2 layers, 100 queries) under fp16 autocast with fp16 flash attention, "
                        "fwd+bwd+Adam, eager launches" % (B, N, k),
            "ms_per_step": round(dt * 1e3, 3), "clouds_per_s": round(B / dt, 2), "steps": steps, "warmup": warmup,
            "loss": float(info["loss"]), "finite": bool(torch.isfinite(info["loss"]))}


def make_step(model, pts, nrm, world=1, lr=1e-3):
    """The training step bench.py times, as one closure (tests/test_step_parity_gpu.py runs exactly this):
    zero_grad -> arena.begin_step -> one multi-tensor bf16 weight cast -> forward under bf16 autocast -> synthetic
    objective -> backward -> gradient packing (+ RCCL all-reduce when world > 1) -> flat Adam.
    Returns dict(step, fwd_bwd, dp, opt, arena, casts)."""
    from gcanet_amd import parallel
    from gcanet_amd.layers import CastCache, ZeroArena
    from gcanet_amd.optim import FlatAdam
    dev = pts.device
    dp = parallel.FlatGradDP(model, world, late=model.encoder.parameters())   # heads' all-reduce overlaps the encoder's backward
    dp.sync_params()
    # option_new.py:83-90 trains with Adam(lr=1e-3): the same rule as ONE elementwise kernel over the flat parameter /
    # gradient / moment buffers (gcanet_amd/optim.py; torch's multi-tensor launches take 0.2 ms for these 57 tensors)
    opt = FlatAdam(dp, lr=lr)
    arena = ZeroArena(dev)            # the small accumulators of a step come pre-zeroed from one allocation: one fill per step
    # bf16 weight copies (in the GEMM kernel's padded operand layout): one multi-tensor cast per step, not one per layer
    casts = CastCache(model, pad_k={model.conv3.weight: (model.conv3.weight.shape[1] + 15) // 16 * 16})

    def fwd_bwd():
        dp.zero_grad()
        arena.begin_step()
        casts.refresh()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(pts, nrm)
        loss = loss_of(out)
        loss.backward()
        return loss

    def step():
        loss = fwd_bwd()
        dp.all_reduce_grads()
        opt.step()
        return loss

    return dict(step=step, fwd_bwd=fwd_bwd, dp=dp, opt=opt, arena=arena, casts=casts)


def capture_step(step, warmup):
    """Warm `step` up on a side stream (as graph capture wants it), capture ONE call into a HIP graph and return
    (graph, loss_buffer); graph.replay() then is one step."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warmup):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    # With a process group alive its watchdog thread polls the events of earlier collectives at any moment; under the
    # default "global" capture mode such a query from another thread during the capture is an error that takes the
    # process down (seen once in three runs of tests/test_parallel_gpu.py).  "thread_local" restricts the check to the
    # capturing thread.
    mode = "thread_local" if (torch.distributed.is_available() and torch.distributed.is_initialized()) else "global"
    with torch.cuda.graph(graph, capture_error_mode=mode):
        loss = step()                                     # not executed: recorded; `loss` is the graph's output buffer
    return graph, loss


def _event_ms(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def north_star_rooflines(dev, B=8, N=8192, k=64, C=128):
    """The two rooflines BASELINE.json's north_star names, at its shape (B=8, N=8192, k=64, C=128), timed with HIP
    events on the launch stream OUTSIDE the timed step (this model's layers are 6/64/64 channels wide, so the C=128
    shape does not occur inside the step):
      knn_gather : KNN(k)(xyz, xyz) [KNN_CUDA signature] + grouping_operation of C=128 features with those lists,
                   against the HBM roofline; algorithmic bytes per SURVEY 8d:
                   kNN 4*B*3*N + 8*B*N*k, group 4*B*C*N + 4*B*N*k + 4*B*C*N*k (= 2.198 GB).
      grouped_mlp: the EdgeConv grouped (N*k, 2C) x (2C, C) contraction, training variant (routed extreme + arg),
                   against the dense bf16 MFMA peak; algorithmic FLOPs 2*(B*N*k)*2C*C = 274.9 G.  The kernel pair
                   executes half of them on the matrix cores (the centre half is contracted once per point)."""
    from gcanet_amd import _lib
    from gcanet_amd.knn_cuda import KNN
    from gcanet_amd.pointnet2_ops.pointnet2_utils import grouping_operation
    g = torch.Generator().manual_seed(7)
    xyz = torch.rand(B, 3, N, generator=g).to(dev)
    feat = torch.randn(B, C, N, generator=g).to(dev)
    knn_mod = KNN(k, transpose_mode=False)
    with torch.no_grad():
        _, I = knn_mod(xyz, xyz)
        idx32 = I.permute(0, 2, 1).to(torch.int32).contiguous()
        ms_knn = _event_ms(lambda: knn_mod(xyz, xyz))
        ms_grp = _event_ms(lambda: grouping_operation(feat, idx32))
    by_knn = 4.0 * B * 3 * N + 8.0 * B * N * k
    by_grp = 4.0 * B * C * N + 4.0 * B * N * k + 4.0 * B * C * N * k
    gbs = (by_knn + by_grp) / (ms_knn + ms_grp) / 1e6
    knn_gather = {"shape": "B=%d,N=%d,k=%d,C=%d" % (B, N, k, C), "knn_ms": round(ms_knn, 4), "group_ms": round(ms_grp, 4),
                  "bytes": by_knn + by_grp, "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                  "frac": round(gbs / PEAK_HBM_GBS, 4), "group_only_frac": round(by_grp / ms_grp / 1e6 / PEAK_HBM_GBS, 4)}
    # grouped MLP at C = Cout = 128
    lib = _lib.lib()
    x = torch.randn(B, N, C, generator=g).to(dev)
    w = (torch.randn(C, 2 * C, generator=g) / (2 * C) ** 0.5).to(dev)
    ga = torch.randn(C, generator=g).to(dev)
    Cp = lib.gcn_edgeconv_padded_channels(C)
    x_bf = torch.empty(B, N, Cp, dtype=torch.bfloat16, device=dev)
    wp = torch.empty(C, 2 * Cp, dtype=torch.bfloat16, device=dev)
    q = torch.empty(B * N, C, device=dev)
    ymax = torch.empty(B, N, C, device=dev)
    amax = torch.empty(B, N, C, dtype=torch.uint8, device=dev)
    gsum = torch.empty(B, 2, 2, dtype=torch.float64, device=dev)
    st = _lib.stream_of(x)
    _lib.call("gcn_cast_pad_bf16", _lib.ptr(x), B * N, C, _lib.ptr(x_bf), st)
    _lib.call("gcn_edgeconv_pack_w", _lib.ptr(w), C, C, _lib.ptr(wp), st)
    ms_c = _event_ms(lambda: _lib.call("gcn_edgeconv_center", _lib.ptr(x_bf), _lib.ptr(wp), B * N, C, C, _lib.ptr(q), st))
    ms_f = _event_ms(lambda: _lib.call("gcn_edgeconv_fwd", _lib.ptr(x_bf), _lib.ptr(wp), _lib.ptr(I), 1, B, N, N, C, k, C, 2,
                                       _lib.ptr(q), _lib.ptr(ymax), None, _lib.ptr(amax), None, _lib.ptr(gsum), _lib.ptr(ga), st))
    flops = 2.0 * B * N * k * 2 * C * C
    executed = 2.0 * B * N * k * C * C + 2.0 * B * N * C * C
    tf = flops / (ms_c + ms_f) / 1e9
    grouped_mlp = {"shape": "B=%d,N=%d,k=%d,C=%d->%d, routed+arg (training)" % (B, N, k, C, C), "flops": flops,
                   "center_ms": round(ms_c, 4), "grouped_ms": round(ms_f, 4), "achieved": round(tf, 1),
                   "peak": PEAK_TFLOPS["bf16_mfma"], "unit": "TFLOP/s", "frac": round(tf / PEAK_TFLOPS["bf16_mfma"], 4),
                   "executed_tflops": round(executed / (ms_c + ms_f) / 1e9, 1),
                   "executed_frac": round(executed / (ms_c + ms_f) / 1e9 / PEAK_TFLOPS["bf16_mfma"], 4)}
    # matrix-pipe busy fraction of the same kernel from the committed PMC passes (SQ_VALU_MFMA_BUSY_CYCLES over the active
    # cycles of the 1024 SIMDs; tools/pmc_mfma.py): clock independent, where `executed_frac` prices against 2.4 GHz
    mp = os.path.join(ROOT, "profiles", "r03_pmc_mfma.json")
    if os.path.exists(mp) and (B, N, k, C) == (8, 8192, 64, 128):
        rec = json.load(open(mp))["kernels"].get("gcn::edgeconv_fwd_q_kernel<8, 4, 2, true, true, true, false>")
        if rec:
            grouped_mlp["mfma_busy_pmc"] = rec["mfma_busy"]
    return {"knn_gather": knn_gather, "grouped_mlp": grouped_mlp}


GROUPING_CFG = dict(similarity_threshold_inst=0.0, min_npoint=30)     # what lets random-init predictions form proposals


def grouping_times(k, B, N, dev, reps=5):
    """forward_grouping (M4:737, the stage right after the timed step) on a batch of BLOB
    clouds (bench.blob_clouds) with the predictions a TRAINED network would hand it -- per-point class scores peaked at
    the blob's class, offsets pointing at the blob's centre, embedding and parameter vectors constant per blob plus
    noise -- so that proposals exist (a random-init network proposes nothing, and a comparison of two empty lists says
    nothing): the fused device path vs the literal per-(cloud, class) path, which must return identical proposals.
    Reported beside the headline number, never part of it."""
    from gcanet_amd.grouping import forward_grouping, forward_grouping_device
    pts, nrm, lab, ctr = blob_clouds(range(B), N, dev, with_centres=True)
    blobs = ctr.shape[1]
    g = torch.Generator().manual_seed(99)
    cls = (lab % 9 + 1).reshape(-1)                                             # foreground classes 1..9
    scores = torch.full((B * N, 10), -4.0, device=dev)
    scores.scatter_(1, cls.view(-1, 1), 4.0)
    offs = (torch.gather(ctr, 1, lab.unsqueeze(-1).expand(-1, -1, 3)) - pts).reshape(-1, 3)
    emb_c = torch.randn(B, blobs, 64, generator=g).to(dev)
    par_c = torch.randn(B, blobs, 22, generator=g).to(dev)
    emb = torch.gather(emb_c, 1, lab.unsqueeze(-1).expand(-1, -1, 64)) + 0.01 * torch.randn(B, N, 64, generator=g).to(dev)
    par = torch.gather(par_c, 1, lab.unsqueeze(-1).expand(-1, -1, 22)) + 0.01 * torch.randn(B, N, 22, generator=g).to(dev)
    args = (scores, offs.contiguous(), torch.arange(B, device=dev).repeat_interleave(N), pts.reshape(-1, 3),
            torch.log_softmax(scores, 1).view(B, N, 10), par, emb)
    cfg = dict(min_npoint=30)
    res = {}
    for name, fn, r in (("device_ms", forward_grouping_device, reps), ("literal_ms", forward_grouping, 2)):
        pi, po = fn(*args, **cfg)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(r):
            pi, po = fn(*args, **cfg)
        torch.cuda.synchronize()
        res[name] = round((time.perf_counter() - t0) / r * 1e3, 3)
        res.setdefault("proposals", int(po.numel()) - 1 if po.numel() else 0)
        res.setdefault("members", int(pi.shape[0]))
        if name == "device_ms":
            dev_out = (pi, po)
        else:
            res["identical"] = bool(torch.equal(dev_out[0], pi) and torch.equal(dev_out[1], po))
    res["note"] = ("forward_grouping on %d blob clouds (%d blobs each) with trained-like predictions (class scores peaked at the "
                   "blob's class, offsets to the blob centre, per-blob embedding / parameter vectors + noise), reference "
                   "thresholds except min_npoint=30; outside the timed step" % (B, blobs))
    return res


def rehearse_cpu(args):
    """Launcher + FlatGradDP + JSON-contract rehearsal over gloo on a box without a GPU (tests/test_parallel_cpu.py).
    The module is a tiny torch stand-in with the hot path's segment structure (an upstream "encoder" whose gradients
    arrive last + downstream "heads"); it is NOT the product path and the line says so."""
    import torch.nn as nn
    from gcanet_amd import parallel
    rank, local, world = parallel.init_distributed("gloo")
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    torch.manual_seed(0)
    enc = nn.Sequential(nn.Linear(6, 32), nn.ReLU())
    heads = nn.Sequential(nn.Linear(32, 64), nn.ReLU(), nn.Linear(64, 10))
    model = nn.Sequential(enc, heads)
    dp = parallel.FlatGradDP(model, world, late=enc.parameters())
    dp.sync_params()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    B, N = args.batch, args.points
    pts, nrm = synth_clouds(range(rank * B, rank * B + B), N, "cpu")
    x = torch.cat([pts, nrm], -1)

    def step():
        dp.zero_grad()
        loss = model(x).pow(2).mean()
        loss.backward()
        dp.all_reduce_grads()
        opt.step()
        return loss

    for _ in range(args.warmup):
        step()
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
        flat = dp.flat.clone()
        torch.distributed.all_reduce(flat, op=torch.distributed.ReduceOp.MAX)
        assert torch.equal(flat, dp.flat), "replicas hold different reduced gradients"
    if rank == 0:
        print(json.dumps({
            "metric": "point-clouds/sec fwd+bwd (N=%d,k=%d)" % (N, args.k), "value": round(world * B * args.steps / dt, 3),
            "unit": "clouds/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "REHEARSAL (cpu stand-in module, gloo) -- not a measurement",
            "config": {"workload": "launcher rehearsal", "global_batch": world * B, "points": N, "k": args.k,
                       "parallelism": "dp%d" % world, "allreduce_overlapped_steps": dp.early_started_in_backward}}))
    if world > 1:
        torch.distributed.destroy_process_group()


def main():
    args = parse_args()
    if args.rehearse_cpu:
        return rehearse_cpu(args)

    from gcanet_amd import _lib, dgcnn, parallel
    rank, local, world = parallel.init_distributed()
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    ndev = torch.cuda.device_count()
    if world > ndev and not args.oversubscribe:
        raise SystemExit("bench.py: %d ranks but only %d GPUs visible (one rank per GPU; --oversubscribe only for a "
                         "rehearsal)" % (world, ndev))
    local = local % ndev
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    _lib.lib()  # fail loudly if the HIP library is missing

    torch.manual_seed(0)
    model = dgcnn.PrimitivesEmbeddingDGCNGn(nn_nb=args.k, dtype="bf16").to(dev)
    # One process per GPU.  With a single rank the whole step (forward, backward, gradient packing, Adam) is captured
    # ONCE into a HIP graph after the warm-up and the timed steps are replays: ~390 launches cost the host ~7.5 ms per
    # step otherwise, about as long as the GPU needs to execute them.  Multi-rank runs launch every kernel from the host
    # (the heads' RCCL all-reduce is started from a backward hook and overlaps the encoder's backward); see DESIGN.md
    # section 6 for what was observed with a captured step followed by an eager all-reduce.
    # (tests/test_parallel_gpu.py captures the whole multi-rank step incl. its RCCL all-reduces on a 1-rank group and
    # replays it correctly; GCANET_GRAPH_MULTIRANK=1 opts a multi-rank run into that mode.  It is not the default: the
    # eager multi-rank step is GPU-bound already (6.7 ms of GPU against 6.0 ms of host enqueue), so a replay would gain
    # ~3 % and an N-rank capture cannot be rehearsed on the one-GPU test box.)
    # a capture needs at least one eager step before it (lazy initialisation, allocator warm-up): --warmup 0 runs eager
    use_graph = (world == 1 or os.environ.get("GCANET_GRAPH_MULTIRANK") == "1") and not args.no_graph and args.warmup >= 1
    B, N = args.batch, args.points
    pts, nrm = synth_clouds(range(rank * B, rank * B + B), N, dev)
    st = make_step(model, pts, nrm, world)
    dp, step = st["dp"], st["step"]

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    run_step = step
    if use_graph:
        # the W untimed warm-up steps: all but the last (the last two from W = 6 on) are launched eagerly before the
        # capture, the rest are REPLAYS of the captured graph -- the first replay after a capture runs ~5 % slow (cold
        # instruction caches / clocks after the synchronising capture: tools/debug/replay_jitter.py) and is warm-up, not work
        replay_warm = 0 if args.warmup < 2 else (1 if args.warmup < 6 else 2)
        graph, loss = capture_step(step, args.warmup - replay_warm)
        run_step = graph.replay
        for _ in range(replay_warm):
            graph.replay()
    else:
        for _ in range(args.warmup):
            step()
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out_loss = run_step()
    t_enq = time.perf_counter() - t0          # host time to ENQUEUE the steps (close to dt = the host is the bottleneck)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if not use_graph:
        loss = out_loss
    # per-kernel durations: the same step, launched from the host with HIP events around every entry point, right after
    # the timed region (a replayed graph has no host-side call to bracket; on the eager path the event records would
    # ride inside the timed steps)
    timed_steps = 3
    if use_graph and hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)   # the capture ran on a side stream
    _lib.enable_timing(True)
    for _ in range(timed_steps):
        step()
    torch.cuda.synchronize()
    timing = _lib.timing_results()
    _lib.enable_timing(False)
    coll = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
        # the collective on its own: the flat fp32 gradient, all-reduced as the step does it (untimed region)
        buf = torch.zeros_like(dp.flat)
        for _ in range(2):
            torch.distributed.all_reduce(buf)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            torch.distributed.all_reduce(buf)
        e1.record()
        torch.cuda.synchronize()
        coll = {"rccl_ranks": torch.distributed.get_world_size(), "backend": torch.distributed.get_backend(),
                "allreduce_bytes": int(buf.numel() * 4), "allreduce_ms": round(e0.elapsed_time(e1) / 5, 4),
                "segments": "heads %d B (async, from inside backward) + encoder %d B" % (dp.split * 4, (buf.numel() - dp.split) * 4)}
    if rank != 0:
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    clouds_per_s = world * B * args.steps / dt
    # dominant hand-written kernel of the step = largest total device time among the timed entry points
    dom = max(timing.items(), key=lambda kv: kv[1][1]) if timing else None
    roofline = None
    kernels = {}
    for tag, (n, tot) in sorted(timing.items(), key=lambda kv: -kv[1][1]):
        km = kernel_model(tag)
        avg = tot / n
        kernels[tag] = {"launches_per_step": n / timed_steps, "avg_ms": round(avg, 4),
                        "tflops": round(km["flops"] / avg / 1e9, 2) if km else None}
    if dom is not None:
        tag, (n, tot) = dom
        km = kernel_model(tag)
        avg_ms = tot / n
        ach = km["flops"] / avg_ms / 1e9
        roofline = {"kernel": tag, "bound": km["bound"], "achieved": round(ach, 2), "peak": km["peak"],
                    "unit": "TFLOP/s", "frac": round(ach / km["peak"], 4), "traffic": pmc_traffic(tag),
                    "avg_launch_ms": round(avg_ms, 4), "launches_per_step": n / timed_steps}
        if "executed_flops" in km:       # the matrix-core work the entry point really issues, against ITS peak
            ex = km["executed_flops"] / avg_ms / 1e9
            roofline.update(executed_tflops=round(ex, 1), executed_peak=km["executed_peak"],
                            executed_frac=round(ex / km["executed_peak"], 4))
        if tag.startswith("knn_model"):
            # since round 2 the N^2 part of this entry point is a FILTER (bf16 MFMA for feature space, packed f32 VALU
            # for xyz+normal) and only ~3k survivors per query get the exact f32 arithmetic: `achieved` stays the
            # algorithmic 2*B*N^2*C f32 FLOPs of SURVEY 8d over the entry point's whole duration (all its kernels)
            roofline["note"] = ("`achieved`/`frac`: ALGORITHMIC exact-f32 distance FLOPs (SURVEY 8d) / time of the whole entry "
                                "point (prep + threshold + filter + exact re-rank kernels) against the f32 peak -- an "
                                "algorithmic-equivalent speed, not a utilisation (it can exceed 1); the N^2 pass runs as a "
                                "bf16-MFMA filter: `executed_*` prices that against the bf16 peak")
    knn_ms = sum(tot for tag, (n, tot) in timing.items() if tag.startswith("knn_model")) / timed_steps
    res = {
        "metric": "point-clouds/sec fwd+bwd (N=%d,k=%d)" % (N, args.k), "value": round(clouds_per_s, 3),
        "unit": "clouds/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "host_enqueue_ms_per_step": round(t_enq / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "step_launch": "hipgraph replay (captured once after warm-up)" if use_graph else "eager",
        "config": {"workload": "BASELINE configs[1]: %d clouds/GPU, N=%d, k=%d, GCANet hot path (DGCNN encoder 3x"
                               "[kNN+EdgeConv], heads, normal EdgeConv, embedding, offset module; M4:634-747) fwd+bwd"
                               "+Adam; stops before forward_grouping/spconv (third-party, SURVEY 8f)" % (B, N, args.k),
                   "global_batch": world * B, "points": N, "k": args.k, "parallelism": "dp%d" % world,
                   "allreduce_overlapped_steps": dp.early_started_in_backward},
        "knn_mpts_per_s": round(3 * B * N / knn_ms / 1e3, 2) if knn_ms > 0 else None,
        "roofline": roofline, "kernels": kernels, "loss": float(loss.detach()),
    }
    if coll is not None:
        res["collective"] = coll
    def guarded(fn):
        """The secondary legs are reported beside the headline and must never cost it: a Python error in one of them
        becomes an `error` entry of the line."""
        try:
            return fn()
        except Exception as e:          # noqa: BLE001
            import traceback
            traceback.print_exc(file=sys.stderr)
            return {"error": "%s: %s" % (type(e).__name__, e)}

    if world == 1:
        res["north_star"] = guarded(lambda: north_star_rooflines(dev))
    if world == 1 and not args.no_full:
        res["full_workload"] = guarded(lambda: full_workload(args, dev))
        if isinstance(res["full_workload"], dict) and "ms_per_step" in res["full_workload"]:
            res["full_workload"]["roofline"] = full_workload_roofline()
        res["cfg5_workload"] = guarded(lambda: cfg5_workload(dev))
    if world == 1 and not args.no_cpu_baseline:
        res["forward_grouping"] = guarded(lambda: grouping_times(args.k, B, N, dev))
        res["cpu_baseline"] = guarded(lambda: cpu_baseline(N, args.k))
    print(json.dumps(res))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
