"""Where does the f32 noise of the KPAM weight gradient come from?  Device f32 step vs the oracle in f64 (truth) and
f32.  Result (round 3): the device is CLOSER to the f64 truth than the f32 oracle is (3.9e-4 vs 4.8e-4 of the tensor's max).
for attention.conv1.2.weight; computing X or the whole key-edge backward in f64 changes nothing)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from oracle import ref_model as R
from gcanet_amd import dgcnn
N, K = 8192, 64
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = dgcnn.PrimitivesEmbeddingDGCNGn(nn_nb=K, dtype="f32")
with torch.no_grad():
    for n_, p_ in m.named_parameters():
        if n_.endswith("weight") and p_.dim() == 1:
            p_.copy_(torch.randn_like(p_))
sd = {k: v.clone() for k, v in m.state_dict().items()}
m = m.to(dev)
pts, nrm = bench.synth_clouds([0], N, "cpu")
names = ["offset_pred_block.attention.conv1.0.weight", "offset_pred_block.attention.conv1.2.weight", "bn1.bias",
         "offset_pred_block.conv1.0.weight", "mlp_seg_prob2.weight", "encoder.conv1.0.weight"]
res = {}
for mode in ("",):
    m.zero_grad(set_to_none=True)
    out = m(pts.to(dev), nrm.to(dev))
    bench.loss_of(out).backward()
    res[mode] = {n_: p_.grad.detach().double().cpu() for n_, p_ in m.named_parameters() if p_.grad is not None}
idxs = [i.cpu() for i in m.encoder.last_idx]
sel = m.offset_pred_block.last_topk_idx.cpu()
def orc(dt):
    leaves = {n: (v.to(dt).clone().requires_grad_(True) if v.dtype.is_floating_point else v) for n, v in sd.items()}
    out, _ = R.hot_path(leaves, pts.to(dt), nrm.to(dt), K, idxs=idxs, topk_idx=sel)
    sum(v.pow(2).mean() for v in out.values()).backward()
    return {n: v.grad.double() for n, v in leaves.items() if getattr(v, "grad", None) is not None}
g64 = orc(torch.float64)
g32 = orc(torch.float32)
rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
for n in names:
    print(n, "oracle f32: %.1e" % rel(g32[n], g64[n]), " ".join("%s: %.1e" % (mo or "device", rel(res[mo][n].reshape(g64[n].shape), g64[n])) for mo in res))
worst = sorted(((rel(res[""][n].reshape(g64[n].shape), g64[n]), rel(g32[n], g64[n]), n) for n in g64), reverse=True)[:8]
print("worst device-vs-f64 (device, oracle f32, name):", [("%.1e" % a, "%.1e" % b, n) for a, b, n in worst])
