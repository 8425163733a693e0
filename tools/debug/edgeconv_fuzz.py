"""Random shapes through the matrix-core EdgeConv kernels (bf16 and IEEE-half operands) against the exact f32 kernel of
the same library on operands that are representable in the 16-bit type: outputs, raw extremes and GroupNorm sums within
1e-4 (f32 summation order is the only difference); k up to 128, ragged N, 6-256 input channels."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gcanet_amd import dgcnn  # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
bad = 0
for it in range(cases):
    B = int(rng.integers(1, 4))
    C = int(rng.choice([6, 16, 32, 40, 64, 100, 128, 256]))
    Cout = 128 if C > 128 else int(rng.choice([64, 128]))
    G = int(rng.choice([2, 4])) if Cout == 128 else 2
    N = int(rng.choice([rng.integers(33, 400), rng.integers(400, 3000), 1024]))
    k = int(min(N, rng.choice([rng.integers(1, 33), 20, 64, 80, 128])))
    if 4 * (k * 2 * C + 2) + 16 * Cout > 150 * 1024:              # the exact f32 kernel keeps a point's k x 2C edge rows in LDS
        k = max(1, (150 * 1024 - 16 * Cout) // (8 * C) - 1)
    prec = str(rng.choice(["bf16", "f16"]))
    t16 = torch.bfloat16 if (prec == "bf16" or C <= 32) else torch.float16
    rnd = lambda t: t.to(t16).float()
    x = rnd(torch.randn(B, C, N, generator=g) * float(10.0 ** rng.uniform(-1, 1)))
    w = torch.randn(Cout, 2 * C, generator=g) / (2 * C) ** 0.5
    w1, wd = rnd(w[:, :C]), rnd(w[:, C:] - w[:, :C])
    w = torch.cat([w1, wd + w1], 1)
    if not torch.equal(rnd(w[:, C:] - w[:, :C]), wd):            # W2 - W1 must come out representable again
        continue
    gamma, beta = torch.randn(Cout, generator=g), torch.randn(Cout, generator=g) * 0.1
    idx = torch.stack([torch.stack([torch.randperm(N, generator=g)[:k] for _ in range(N)]) for _ in range(B)]) if N <= 400 \
        else torch.randint(0, N, (B, N, k), generator=g)
    args = [t.to(dev) for t in (x, idx, w, gamma, beta)]
    a = dgcnn.edgeconv_forward_raw(*args, G, prec)
    b = dgcnn.edgeconv_forward_raw(*args, G, "f32")
    ok = True
    for name in ("out", "ymax", "ymin"):
        sc = float(b[name].abs().max()) + 1e-6
        err = float((a[name] - b[name]).abs().max()) / sc
        if not err <= 1e-4:
            ok = False
            print("MISMATCH case %d: B=%d C=%d N=%d k=%d Cout=%d G=%d %s: %s rel err %.3g" % (it, B, C, N, k, Cout, G, prec, name, err), flush=True)
            break
    gs = float((a["gsum"] - b["gsum"]).abs().max() / (b["gsum"].abs().max() + 1e-9))
    if ok and not gs <= 1e-4:
        ok = False
        print("MISMATCH case %d: gsum rel err %.3g (C=%d N=%d k=%d %s)" % (it, gs, C, N, k, prec), flush=True)
    bad += 0 if ok else 1
print("cases %d, mismatches %d" % (cases, bad))
