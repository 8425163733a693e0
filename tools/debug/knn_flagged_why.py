"""Why does csrc/knn_filter.hip flag a query?  Reads the call's workspace (layout of knnf_layout) and classifies the
flagged queries: too few candidates, too many, or proof failed (with the margins)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gcanet_amd import _lib, dgcnn
C, N, B, k = 64, 8192, 8, 64
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(C + N)
kind = sys.argv[1] if len(sys.argv) > 1 else "normal"
x = torch.randn(B, N, C, generator=g)
if kind == "blobs":
    x = torch.randn(B, 64, C, generator=g)[:, torch.arange(N) % 64] + 0.03 * x
x = x.to(dev)
lib = _lib.lib()
ws = torch.empty(lib.gcn_knn_feature_ws_bytes(B, N, C), dtype=torch.uint8, device=dev)
idx = torch.empty(B, N, k, dtype=torch.int64, device=dev)
_lib.call("gcn_knn_feature", _lib.ptr(x), B, N, C, k, k, _lib.ptr(idx), _lib.ptr(ws), _lib.stream_of(x))
torch.cuda.synchronize()
al = lambda v: (v + 255) & ~255
Cp, Np = C, (N + 127) // 128 * 128
o = 0
off = {}
for name, size in (("msum", 4 * B * Cp), ("stat", 8 * B), ("nflag", 256), ("ut", 2 * B * Np * Cp), ("hn", 4 * B * Np), ("xx", 4 * B * N),
                   ("theta", 4 * B * N), ("tau", 4 * B * N), ("flag", B * N), ("flist", 4 * B * N), ("bitmap", 4 * B * N * (Np // 32)),
                   ("keys", 4 * B * N * 512), ("cjs", 2 * B * N * 512), ("ccnt", 4 * B * N)):
    off[name] = o; o += al(size)
view = lambda name, dt, n: ws[off[name]:off[name] + n * torch.empty((), dtype=dt).element_size()].view(dt)
flag = view("flag", torch.uint8, B * N).view(B, N)
tau = view("tau", torch.float32, B * N).view(B, N)
ccnt = view("ccnt", torch.int32, B * N).view(B, N)
hn = view("hn", torch.float32, B * Np).view(B, Np)
stat = view("stat", torch.float32, 2 * B).view(B, 2)
bm = view("bitmap", torch.int32, B * N * (Np // 32)).view(B, N, Np // 32)
fl = flag.nonzero()
print("flagged", fl.shape[0], "nflag", int(view("nflag", torch.int32, 1)[0]))
for b, q in fl.tolist()[:12]:
    xb = x[b]
    key = ((xb * xb).sum(1) + (xb[q] * xb[q]).sum() - 2 * xb @ xb[q])
    dk = key.kthvalue(k)[0].item()
    bits = sum(bin(w & 0xffffffff).count("1") for w in bm[b, q].tolist())
    nq = 2 * hn[b, q].item(); R2, X2 = stat[b, 0].item(), stat[b, 1].item()
    s = nq ** 0.5 + R2 ** 0.5; sx = (xb[q] * xb[q]).sum().item() ** 0.5 + X2 ** 0.5
    eta = 0.00393 * s; df = 8 * (Cp + 16) * 2 ** -24 * s * s; dr = (C + 3) * 2 ** -24 * sx * sx
    t = tau[b, q].item()
    root = max(t - df, 0) ** 0.5 - eta
    print("b %d q %d: candidates %d (ccnt %d) tau %.4f dk %.4f  L %.4f  eta %.4f Df %.4f Dref %.4f |ut_q| %.3f R %.3f rank(tau) %d"
          % (b, q, bits, ccnt[b, q].item(), t, dk, root * root - dr, eta, df, dr, nq ** 0.5, R2 ** 0.5, int((key <= t).sum())))
