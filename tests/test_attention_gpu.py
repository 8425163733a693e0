"""GPU parity of the attention stacks (gcanet_amd.transformer / query_decoder, fused HIP attention core)
vs golden vectors produced by the reference's own modules (tests/golden/make_golden.py, section 6)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_sdpa_kernel_vs_torch(dev):
    from gcanet_amd.attention import sdpa
    g = torch.Generator().manual_seed(0)
    for (BH, Lq, Lk, D) in ((3, 70, 130, 32), (2, 100, 1000, 8), (4, 65, 64, 64), (1, 1, 5, 16)):
        q, k, v = [torch.randn(BH, L, D, generator=g).to(dev).requires_grad_() for L in (Lq, Lk, Lk)]
        mask = (torch.rand(Lq, Lk, generator=g) < 0.3).to(dev)
        mask[:, 0] = False
        for m in (None, mask):
            out = sdpa(q, k, v, m, 0.2)
            s = torch.bmm(q, k.transpose(1, 2)) * 0.2
            if m is not None:
                s = s.masked_fill(m.unsqueeze(0), float("-inf"))
            ref = torch.bmm(torch.softmax(s, -1), v)
            np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=1e-4, atol=1e-5)
            go = torch.randn_like(out)
            g1 = torch.autograd.grad(out, (q, k, v), go)
            g2 = torch.autograd.grad(ref, (q, k, v), go)
            for a, b in zip(g1, g2):
                np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-3, atol=1e-5)


def test_transformer_matches_reference_golden(dev, att_golden):
    from gcanet_amd.transformer import Transformer
    g = att_golden
    T = Transformer(dim=32, depth=2, heads=4, dim_head=8, mlp_dim=64, dropout=0.0)
    T.load_state_dict({k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("tr_sd_")})
    T = T.to(dev)
    x = torch.from_numpy(g["tr_x"]).to(dev).requires_grad_()
    y = T(x)
    np.testing.assert_allclose(y.detach().cpu().numpy(), g["tr_y"], rtol=1e-4, atol=1e-5)
    (y * torch.from_numpy(g["tr_gy"]).to(dev)).sum().backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), g["tr_dx"], rtol=1e-3, atol=1e-5)
    for n_, p_ in T.named_parameters():
        np.testing.assert_allclose(p_.grad.cpu().numpy(), g["tr_grad_" + n_], rtol=2e-3, atol=2e-5, err_msg=n_)


@pytest.mark.parametrize("tag,kw", [("qd", dict(iter_pred=False, attn_mask=False)),
                                    ("qdi", dict(iter_pred=True, attn_mask=True, pe=True))])
def test_query_decoder_matches_reference_golden(dev, att_golden, tag, kw):
    from gcanet_amd.query_decoder import QueryDecoder
    g = att_golden
    Q = QueryDecoder(num_layer=2, num_query=10, num_class=5, in_channel=16, d_model=32, nhead=4, hidden_dim=64, **kw)
    Q.load_state_dict({k[len(tag) + 4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(tag + "_sd_")})
    Q = Q.to(dev).eval()
    offs = [int(v) for v in g[tag + "_offsets"]]
    with torch.no_grad():
        o = Q(torch.from_numpy(g[tag + "_x"]).to(dev), offs)
    for k_ in ("labels", "scores", "parameters"):
        np.testing.assert_allclose(o[k_].cpu().numpy(), g[tag + "_" + k_], rtol=1e-4, atol=1e-4)
    for i, m in enumerate(o["masks"]):
        np.testing.assert_allclose(m.cpu().numpy(), g[tag + "_mask%d" % i], rtol=1e-4, atol=1e-4)
    if kw["iter_pred"]:
        for li, aux in enumerate(o["aux_outputs"]):
            np.testing.assert_allclose(aux["labels"].cpu().numpy(), g[tag + "_aux%d_labels" % li], rtol=1e-4, atol=1e-4)


def test_transformer_masked_matches_reference_golden(dev):
    """Mask with padded tokens (fully masked query rows attend uniformly in the reference: finite -finfo.max fill,
    transformer.py:57-67), forward and input gradient vs the reference's own module."""
    import os
    from gcanet_amd.transformer import Transformer
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "transformer_mask_golden.npz"))
    T = Transformer(dim=32, depth=2, heads=4, dim_head=8, mlp_dim=64, dropout=0.0)
    T.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd_")})
    T = T.to(dev)
    x = torch.from_numpy(g["x"]).to(dev).requires_grad_()
    y = T(x, mask=torch.from_numpy(g["mask"]).to(dev))
    np.testing.assert_allclose(y.detach().cpu().numpy(), g["y"], rtol=1e-4, atol=1e-5)
    (y * torch.from_numpy(g["gy"]).to(dev)).sum().backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), g["dx"], rtol=1e-4, atol=1e-4 * np.abs(g["dx"]).max())


def test_mha_refuses_attention_dropout_in_training(dev):
    from gcanet_amd.query_decoder import SelfAttentionLayer
    layer = SelfAttentionLayer(d_model=32, nhead=4, dropout=0.1).to(dev).train()
    with pytest.raises(RuntimeError, match="dropout"):
        layer(torch.randn(2, 10, 32, device=dev))
    layer.eval()
    assert torch.isfinite(layer(torch.randn(2, 10, 32, device=dev))).all()


@pytest.mark.parametrize("kw", [dict(iter_pred=False, attn_mask=False), dict(iter_pred=True, attn_mask=True, pe=True)])
@pytest.mark.parametrize("precision", ["f32", "fp16"])
def test_query_decoder_equal_length_batch_equals_per_cloud_loop(dev, kw, precision):
    """Clouds of equal length take ONE attention launch over B*heads (gcanet_amd/query_decoder.py: CrossAttentionLayer);
    the reference's per-cloud loop (query_decoder.py:32-43) is kept for ragged batches.  Same outputs and gradients:
    offsets handed over as a device tensor keep the loop."""
    from gcanet_amd.query_decoder import QueryDecoder
    torch.manual_seed(3)
    B, n, C = 3, 640, 16
    Q = QueryDecoder(num_layer=2, num_query=20, num_class=5, in_channel=C, d_model=64, nhead=2, hidden_dim=64,
                     precision=precision, **kw).to(dev)
    x = torch.randn(B * n, C, device=dev)
    offs = [i * n for i in range(B + 1)]
    outs = []
    for o in (offs, torch.tensor(offs, device=dev)):
        Q.zero_grad()
        xi = x.clone().requires_grad_()
        r = Q(xi, o)
        loss = r["labels"].float().sum() + r["parameters"].float().pow(2).sum() + sum(m.float().mean() for m in r["masks"])
        loss.backward()
        outs.append((r, xi.grad.clone(), {k: p.grad.clone() for k, p in Q.named_parameters() if p.grad is not None}))
    (ra, ga, pa), (rb, gb, pb) = outs
    tol = dict(rtol=1e-5, atol=1e-6) if precision == "f32" else dict(rtol=2e-3, atol=2e-4)
    for k_ in ("labels", "scores", "parameters"):
        np.testing.assert_allclose(ra[k_].detach().cpu().numpy(), rb[k_].detach().cpu().numpy(), **tol)
    for ma, mb in zip(ra["masks"], rb["masks"]):
        np.testing.assert_allclose(ma.detach().cpu().numpy(), mb.detach().cpu().numpy(), **tol)
    np.testing.assert_allclose(ga.cpu().numpy(), gb.cpu().numpy(), rtol=tol["rtol"] * 10, atol=tol["atol"] * 10 * float(gb.abs().max()))
    for k_ in pa:
        np.testing.assert_allclose(pa[k_].cpu().numpy(), pb[k_].cpu().numpy(), rtol=tol["rtol"] * 10,
                                   atol=tol["atol"] * 10 * float(pb[k_].abs().max()) + 1e-7, err_msg=k_)
