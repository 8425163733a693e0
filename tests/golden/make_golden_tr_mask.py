#!/usr/bin/env python3
"""Masked-Transformer golden vectors from the reference's own models/transformer.py (build container only).
The mask marks padded tokens; the reference fills masked scores with the finite -finfo.max, so a padded QUERY token
(a fully masked row) attends uniformly to all keys (transformer.py:57-67) -- the case the r1 goldens did not cover.

Usage:  python tests/golden/make_golden_tr_mask.py
"""
import importlib
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)


def main():
    torch.set_num_threads(1)
    tr = importlib.import_module("models.transformer")
    torch.manual_seed(17)
    g = torch.Generator().manual_seed(99)
    T = tr.Transformer(dim=32, depth=2, heads=4, dim_head=8, mlp_dim=64, dropout=0.0)
    # batch 1: the reference multiplies the (b,n,n) mask into (b,h,n,n) scores without a head axis
    # (transformer.py:62-65), which only broadcasts for b == 1 (b == h would run and pair batch i with head i)
    x = torch.randn(1, 40, 32, generator=g, requires_grad=True)
    mask = torch.ones(1, 39, dtype=torch.bool)              # the reference pads one leading True (cls token)
    mask[0, 30:] = False                                    # padded tail
    mask[0, 5:9] = False                                    # holes
    y = T(x, mask=mask)
    gy = torch.randn(y.shape, generator=g)
    (y * gy).sum().backward()
    out = {"sd_" + k: v.detach().numpy().copy() for k, v in T.state_dict().items()}
    out.update({"x": x.detach().numpy(), "mask": mask.numpy(), "y": y.detach().numpy(), "gy": gy.numpy(), "dx": x.grad.numpy()})
    np.savez_compressed(os.path.join(OUT, "transformer_mask_golden.npz"), **out)
    print("transformer_mask_golden.npz", os.path.getsize(os.path.join(OUT, "transformer_mask_golden.npz")), "bytes")


if __name__ == "__main__":
    main()
