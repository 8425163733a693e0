"""Loss-side callers (gcanet_amd/losses.py) against a literal CPU restatement of utils/loss_utils.py:203-257,308-435
(oracle/ref_model.py).  The reference has no fixtures for them (parity unpinned by reference data).  fp32, 1e-5."""
import numpy as np
import pytest
import torch

from oracle import ref_model as R

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", ["mixed", "single_label_cloud", "with_unlabelled"])
def test_embedding_loss(dev, case):
    from gcanet_amd.losses import compute_embedding_loss
    g = torch.Generator().manual_seed(3)
    B, N, K = 3, 700, 16
    feat = torch.randn(B, N, K, generator=g)
    lab = torch.randint(0, 6, (B, N), generator=g)
    lab[lab == 4] = 5                                        # a label value that never occurs
    if case == "single_label_cloud":
        lab[1] = 2                                           # push term skipped for this cloud
    if case == "with_unlabelled":
        lab[0, :100] = -1
    fd = feat.to(dev).requires_grad_(True)
    loss, pull, push = compute_embedding_loss(fd, lab.to(dev))
    loss.sum().backward()
    fr = feat.clone().requires_grad_(True)
    lr, pr, qr = R.embedding_loss(fr, lab)
    lr.sum().backward()
    assert loss.shape == (1,)
    np.testing.assert_allclose(pull.detach().cpu().numpy(), pr.detach().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(push.detach().cpu().numpy(), qr.detach().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(fd.grad.cpu().numpy(), fr.grad.numpy(), rtol=1e-4, atol=1e-7)


def test_instance_loss(dev):
    from gcanet_amd.losses import instance_loss
    rng = np.random.default_rng(5)
    g = torch.Generator().manual_seed(5)
    Np, nI, P, C = 6000, 17, 30, 10
    inst = rng.integers(0, nI, Np).astype(np.int64)
    inst[rng.random(Np) < 0.1] = -100
    pointnum = np.bincount(inst[inst >= 0], minlength=nI).astype(np.int32)
    cls = rng.integers(0, C - 1, nI).astype(np.int64)
    # proposals: most of them a noisy copy of one instance (so that IoU >= 0.5 happens), some random
    members, offs = [], [0]
    for p in range(P):
        if p % 4 != 3:
            pts = np.nonzero(inst == rng.integers(0, nI))[0]
            pts = np.concatenate([rng.choice(pts, int(0.8 * len(pts)), replace=False), rng.integers(0, Np, 20)])
        else:
            pts = rng.integers(0, Np, 150)
        members.append(pts)
        offs.append(offs[-1] + len(pts))
    S = offs[-1]
    pidx = torch.from_numpy(np.stack([np.repeat(np.arange(P), np.diff(offs)), np.concatenate(members)], 1).astype(np.int32))
    poff = torch.tensor(offs, dtype=torch.int32)
    cls_s, iou_s = torch.randn(P, C, generator=g), torch.randn(P, C, generator=g)
    mask_s = torch.randn(S, C, generator=g)
    ibi = pidx[:, 0].long()
    t = lambda a: torch.from_numpy(a)
    dv = [x.to(dev).requires_grad_(True) for x in (cls_s, mask_s, iou_s)]
    loss = instance_loss(dv[0], dv[1], dv[2], pidx, poff, t(inst).to(dev), t(pointnum).to(dev), t(cls).to(dev), ibi.to(dev), C)
    loss.backward()
    rv = [x.clone().requires_grad_(True) for x in (cls_s, mask_s, iou_s)]
    lossr = R.instance_loss(rv[0], rv[1], rv[2], pidx, poff, t(inst), t(pointnum), t(cls), ibi, C)
    lossr.backward()
    np.testing.assert_allclose(float(loss), float(lossr), rtol=1e-5)
    for a, b in zip(dv, rv):
        np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.numpy(), rtol=1e-4, atol=1e-7)
