"""Debug: EdgeConv bf16 forward vs a torch reference on the GPU, with mismatch patterns."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gcanet_amd import dgcnn

dev = torch.device("cuda:0")
def bf(t): return t.to(torch.bfloat16).float()
for (B, C, N, k, Cout) in [(2, 16, 96, 8, 64), (1, 64, 257, 64, 128), (1, 128, 130, 64, 128), (2, 6, 200, 16, 64), (1, 32, 90, 80, 128), (1, 128, 64, 33, 64)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, C, N, generator=g)
    idx = torch.stack([torch.stack([torch.randperm(N, generator=g)[:k] for _ in range(N)]) for _ in range(B)])
    w = torch.randn(Cout, 2 * C, generator=g) / (2 * C) ** 0.5
    gamma = torch.randn(Cout, generator=g); beta = torch.randn(Cout, generator=g) * 0.1
    r = dgcnn.edgeconv_forward_raw(x.to(dev), idx.to(dev), w.to(dev), gamma.to(dev), beta.to(dev), 2, "bf16", need_arg=True)
    torch.cuda.synchronize()
    xr = bf(x); w1 = bf(w[:, :C]); wd = bf(w[:, C:] - w[:, :C])
    xt = xr.permute(0, 2, 1)                                   # (B,N,C)
    nb = torch.stack([xt[b][idx[b]] for b in range(B)])        # (B,N,k,C)
    y = torch.einsum("bnkc,oc->bnko", nb.double(), w1.double()) + torch.einsum("bnc,oc->bno", xt.double(), wd.double()).unsqueeze(2)
    ymax, ymin = y.max(2)[0], y.min(2)[0]
    gm, gn = r["ymax"].cpu().double(), r["ymin"].cpu().double()
    bad = (gm - ymax).abs() > 1e-4 * (1 + ymax.abs())
    badn = (gn - ymin).abs() > 1e-4 * (1 + ymin.abs())
    print("case", (B, C, N, k, Cout), "ymax bad", int(bad.sum()), "of", bad.numel(), "ymin bad", int(badn.sum()),
          "max err", float((gm - ymax).abs().max()))
    if bad.any():
        bi = bad.nonzero()
        print("  bad points (b,n) sample:", sorted(set((int(a), int(b_)) for a, b_, _ in bi[:200].tolist()))[:20])
        print("  bad cols sample:", sorted(set(int(c) for _, _, c in bi[:400].tolist()))[:40])
        a0, n0_, c0 = bi[0].tolist()
        print("  first bad", (a0, n0_, c0), "got", float(gm[a0, n0_, c0]), "want", float(ymax[a0, n0_, c0]), "row values", y[a0, n0_, :, c0][:8].tolist())
    am = r["amax"].cpu().long()
    print("  amax range", int(am.min()), int(am.max()), "(k=%d)" % k)
    cnt = (Cout // 2) * N * k
    yg = y.permute(0, 3, 1, 2).reshape(B, 2, -1)
    print("  gsum mean got", (r["gsum"].cpu()[..., 0] / cnt).flatten().tolist(), "want", yg.mean(-1).flatten().tolist())
    print("  gsum sq   got", (r["gsum"].cpu()[..., 1] / cnt).flatten().tolist(), "want", (yg * yg).mean(-1).flatten().tolist())
