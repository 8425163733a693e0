"""Offset module of the reference's model variant M2 (models/dgcnn-hais-concat-direct-2.py:326-462): the one place the
reference runs KNN_CUDA + pointnet2 grouping_operation inside the network graph (M2:401-415).  Golden vectors:
tests/golden/m2_offset_golden.npz = the reference's own source text executed on the oracle's native-op restatements
(tests/golden/make_golden_m2.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_model as R

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "m2_offset_golden.npz")


def _gold():
    return np.load(GOLD)


def test_oracle_restatement_matches_m2_golden():
    """CPU: oracle/ref_model.offset_pred_module_m2 against the reference-text output, forward and input gradients."""
    g = _gold()
    sd = {k[3:]: g[k] for k in g.files if k.startswith("sd_")}
    t = lambda k: torch.from_numpy(g[k]).requires_grad_()
    pts, feat, sem, ins = t("points"), t("feat"), t("sem"), t("ins")
    o = R.offset_pred_module_m2(pts, feat, sem, ins, sd)
    np.testing.assert_allclose(o.detach().numpy(), g["out"], rtol=1e-4, atol=5e-5)   # f32 CPU kernels differ by host (summation order)
    (o * torch.from_numpy(g["gout"])).sum().backward()
    for got, key in ((pts.grad, "dpoints"), (feat.grad, "dfeat"), (ins.grad, "dins")):
        np.testing.assert_allclose(got.numpy(), g[key], rtol=1e-4, atol=1e-4 * np.abs(g[key]).max(), err_msg=key)
    assert bool(g["dsem_is_none"]) and sem.grad is None          # semantic distances are computed and unused (M2:444)


@pytest.mark.gpu
def test_m2_offset_module_matches_reference_golden(dev):
    """GPU: gcanet_amd.dgcnn2.OFFSET_PRED_MODULE (HIP kNN + HIP grouping_operation fwd/bwd + fused grouped block)
    with the reference's weights: outputs, input gradients and parameter gradients within 1e-4 (scale-relative for
    gradients, which are sums over N*k terms)."""
    from gcanet_amd import dgcnn2
    g = _gold()
    m = dgcnn2.OFFSET_PRED_MODULE(nn_nb=60, sampling_ratio=120)
    missing, unexpected = m.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd_")}, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    m = m.to(dev)
    t = lambda k: torch.from_numpy(g[k]).to(dev).requires_grad_()
    pts, feat, sem, ins = t("points"), t("feat"), t("sem"), t("ins")
    o = m(pts, feat, sem, ins, None)
    np.testing.assert_allclose(o.detach().cpu().numpy(), g["out"], rtol=1e-4, atol=1e-4)
    (o * torch.from_numpy(g["gout"]).to(dev)).sum().backward()
    for got, key in ((pts.grad, "dpoints"), (feat.grad, "dfeat"), (ins.grad, "dins")):
        np.testing.assert_allclose(got.cpu().numpy(), g[key], rtol=1e-4, atol=1e-4 * np.abs(g[key]).max(), err_msg=key)
    for n_, p_ in m.named_parameters():
        key = "grad_" + n_
        if key in g.files:
            np.testing.assert_allclose(p_.grad.cpu().numpy(), g[key], rtol=1e-4, atol=1e-4 * np.abs(g[key]).max(), err_msg=n_)
        else:
            assert p_.grad is None or float(p_.grad.abs().max()) == 0.0, n_      # attention_seg is never called
