"""Drop-in for ``models/transformer.py`` (ViT-style pre-norm Transformer): same class names, constructor
arguments and parameter names; the attention core runs through the fused HIP kernel instead of the
reference's einsum -> softmax -> einsum that materialises (b,h,n,n) (transformer.py:52-69)."""
import torch
import torch.nn.functional as F
from torch import nn

from .attention import sdpa


class Residual(nn.Module):
    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def forward(self, x, **kwargs):
        return self.fn(x, **kwargs) + x


class PreNorm(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fn = fn

    def forward(self, x, **kwargs):
        return self.fn(self.norm(x), **kwargs)


class FeedForward(nn.Module):
    def __init__(self, dim, hidden_dim, dropout=0.):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(dim, hidden_dim), nn.GELU(), nn.Dropout(dropout),
                                 nn.Linear(hidden_dim, dim), nn.Dropout(dropout))

    def forward(self, x):
        return self.net(x)


class Attention(nn.Module):
    """transformer.py:36-75.  NOTE the reference scales by dim ** -0.5 (the MODEL width, not dim_head)."""

    def __init__(self, dim, heads=8, dim_head=64, dropout=0., precision="f32"):
        super().__init__()
        self.precision = precision      # "f32" exact kernel | "bf16" / "fp16" matrix-core flash kernels (dim_head 32/64)
        inner_dim = dim_head * heads
        self.heads = heads
        self.scale = dim ** -0.5
        self.to_qkv = nn.Linear(dim, inner_dim * 3, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner_dim, dim), nn.Dropout(dropout))

    def forward(self, x, mask=None):
        b, n, _ = x.shape
        h = self.heads
        q, k, v = [t.view(b, n, h, -1).transpose(1, 2).reshape(b * h, n, -1) for t in self.to_qkv(x).chunk(3, dim=-1)]
        am = None
        if mask is not None:   # transformer.py:57-62: pad a leading True, outer product, fill ~mask
            mask = F.pad(mask.flatten(1), (1, 0), value=True)
            assert mask.shape[-1] == n, 'mask has incorrect dimensions'
            keep = mask[:, None, :] * mask[:, :, None]                       # (b,n,n)
            # The reference fills masked scores with the FINITE -finfo.max, so a padded query token (a fully masked
            # row) soft-maxes to the uniform distribution over all n keys, i.e. mean(V) -- not to zeros.  Same
            # result here: such rows are un-masked and their query is zeroed (all scores equal -> uniform).
            dead = ~mask                                                     # (b,n) padded query tokens
            keep = keep | dead[:, :, None]
            q = q.masked_fill(dead.repeat_interleave(h, 0).unsqueeze(-1), 0.0)
            am = (~keep).unsqueeze(1).expand(-1, h, -1, -1).reshape(b * h, n, n)
        out = sdpa(q, k, v, am, self.scale, self.precision)
        out = out.view(b, h, n, -1).transpose(1, 2).reshape(b, n, -1)
        return self.to_out(out)


class Transformer(nn.Module):
    def __init__(self, dim, depth, heads, dim_head, mlp_dim, dropout, precision="f32"):
        super().__init__()
        self.layers = nn.ModuleList([])
        for _ in range(depth):
            self.layers.append(nn.ModuleList([
                Residual(PreNorm(dim, Attention(dim, heads=heads, dim_head=dim_head, dropout=dropout, precision=precision))),
                Residual(PreNorm(dim, FeedForward(dim, mlp_dim, dropout=dropout)))]))

    def forward(self, x, mask=None):
        for attn, ff in self.layers:
            x = attn(x, mask=mask)
            x = ff(x)
        return x
