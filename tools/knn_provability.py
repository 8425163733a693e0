"""Which lever makes the feature-space kNN prefilter (csrc/knn_filter.hip) hold on a given feature distribution?
For the encoder's actual layer-2/3 inputs (uniform and blob clouds) and the synthetic cases of knn_fallback_bench.py,
per query: the smallest threshold tau_need at which the a-posteriori proof can succeed given the exact k-th key, and
how many candidates lie under it -- with (a) the round-2 constants, (b) the corrected bf16 rounding bound (2^-8),
(c) a tighter bound on the reference's own f32 evaluation error, (d) a two-term bf16 split (16 significant bits).
A query is servable by the filter iff that count is <= CAP."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gcanet_amd import dgcnn
dev = torch.device("cuda:0")
K, CAP = 64, 512

def analyse(name, x):            # x (N,C) f32 on the device
    N, C = x.shape
    xd = x.double()
    u = x - x.mean(0, keepdim=True)
    def bf(t): return t.to(torch.bfloat16).float()
    hi = bf(u); lo = bf(u - hi)
    xx = (x * x).sum(1)
    key = (xx[:, None] + xx[None, :] - 2 * (x @ x.t()))                       # ~ the reference's f32 key
    dk = key.kthvalue(K, dim=1)[0]
    X = xx.max().sqrt(); xq = xx.sqrt()
    out = []
    for label, img, rel in (("bf16", hi, 2.0 ** -8), ("split", hi + lo, 2.0 ** -16)):
        n2 = (img * img).sum(1)
        a = (n2[:, None] + n2[None, :] - 2 * (img.double() @ img.double().t()).float()).clamp_min(0)
        nq = n2.sqrt(); R = nq.max()
        for dlabel, dfac in (("loose", 4.0 * (C + 8)), ("tight", 1.0 * (C + 4))):
            for elabel, efac in ((("r2", 0.00198 / 2.0 ** -8),) if label == "bf16" else ()) + (("ok", 1.003),):
                eta = efac * rel * (nq + R)
                big = torch.maximum(nq, xq) + torch.maximum(R, X)
                Delta = dfac * 2.0 ** -24 * big * big
                need = (torch.sqrt(dk.clamp_min(0) + Delta) + eta) ** 2 * 1.0001 + Delta
                cnt = (a <= need[:, None]).sum(1)
                ok = ((cnt <= CAP) & (cnt >= K)).float().mean().item()
                out.append("%s/%s/%s: servable %.3f (median cand %d, p90 %d)" % (label, dlabel, elabel, ok, int(cnt.median()), int(cnt.float().quantile(0.9))))
    print(name, "| N=%d C=%d dk median %.3g, |u| median %.3g, X %.3g" % (N, C, dk.median().item(), u.norm(dim=1).median().item(), X.item()))
    for o in out: print("    ", o)

torch.manual_seed(0)
m = dgcnn.PrimitivesEmbeddingDGCNGn(nn_nb=K, dtype="bf16").to(dev)
m.encoder.keep_feats = True
for cname in ("uniform", "blobs"):
    if cname == "uniform":
        pts, nrm = bench.synth_clouds([0, 1], 8192, dev)
    else:
        pts, nrm, _ = bench.blob_clouds([0, 1], 8192, dev)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        m(pts, nrm)
    for li, f in enumerate(m.encoder.last_feats):
        analyse("%s cloud, layer-%d input" % (cname, li + 2), f[0].float())
g = torch.Generator().manual_seed(1)
analyse("synthetic flat patches", (torch.randn(64, 64, generator=g)[torch.arange(8192) % 64] + 1e-4 * torch.randn(8192, 64, generator=g)).to(dev))
analyse("synthetic offset", (torch.randn(8192, 64, generator=g) * 0.05 + 4.0).to(dev))
analyse("synthetic uniform", torch.randn(8192, 64, generator=g).to(dev))
