"""Which backward kernel is not reproducible?  fwd_bwd is run R times with fixed parameters; the gradient arriving at
the output of EVERY custom autograd.Function (= the input of its backward) and every parameter gradient is recorded and
compared with run 0.  The culprit is the op whose incoming gradients are identical while something upstream of it is not.
usage: bwd_determinism.py B R"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from gcanet_amd import dgcnn, layers, losses
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
R = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = dgcnn.PrimitivesEmbeddingDGCNGn(nn_nb=64, dtype="bf16").to(dev)
pts, nrm = bench.synth_clouds(range(B), 8192, dev)
st = bench.make_step(m, pts, nrm, world=1)
rec, order = {}, []

def wrap(cls):
    orig = cls.apply
    def apply(*a, **kw):
        out = orig(*a, **kw)
        outs = out if isinstance(out, tuple) else (out,)
        ci = sum(1 for k in order if k[0] == cls.__name__)
        order.append((cls.__name__, ci))
        for oi, o in enumerate(outs):
            if torch.is_tensor(o) and o.requires_grad:
                key = "%02d %s#%d.out%d %s" % (len(order), cls.__name__, ci, oi, tuple(o.shape))
                o.register_hook(lambda g, key=key: rec.__setitem__(key, g.detach().float().clone()))
        return out
    cls.apply = apply

for mod in (dgcnn, layers, losses):
    for name in dir(mod):
        c = getattr(mod, name)
        if isinstance(c, type) and issubclass(c, torch.autograd.Function) and c is not torch.autograd.Function and c.__module__ == mod.__name__:
            wrap(c)

runs = []
for r in range(R):
    rec.clear(); order.clear()
    st["fwd_bwd"](); st["dp"].pack_grads()
    torch.cuda.synchronize()
    cur = dict(rec)
    off = 0
    for n_, p_ in m.named_parameters():
        cur["param " + n_] = st["dp"].flat[off:off + p_.numel()].clone(); off += p_.numel()
    runs.append(cur)
for r in range(1, R):
    bad = []
    for k in runs[0]:
        a, b = runs[0][k], runs[r][k]
        if not torch.equal(a, b):
            bad.append("%s %.1e" % (k, float((a - b).abs().max()) / max(float(a.abs().max()), 1e-30)))
    print("run %d vs 0: %d of %d differ" % (r, len(bad), len(runs[0])))
    for s in bad: print("    ", s)
    sys.stdout.flush()
