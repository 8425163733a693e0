"""Attention core shared by gcanet_amd.transformer and gcanet_amd.query_decoder.
precision "f32": exact fused forward (csrc/attention.hip, parity path); its backward recomputes the probabilities
from the saved log-sum-exp in torch ops (small sequences only).  precision "bf16": matrix-core flash kernels,
forward and backward (csrc/attention_mfma.hip) -- nothing of size Lq x Lk is ever materialised."""
import torch

from . import _lib


def _workspace(BH, Lq, Lk, D, device):
    n = _lib.lib().gcn_attention_ws_bytes(BH, Lq, Lk, D)
    if n < 0:
        raise RuntimeError("gcn_attention_ws_bytes: bad shape")
    return torch.empty(n, dtype=torch.uint8, device=device)


class SDPAFunction(torch.autograd.Function):
    """out = softmax(scale * q k^T [+ mask]) v  for q (BH,Lq,D), k/v (BH,Lk,D); mask: bool, True = masked out."""

    @staticmethod
    def forward(ctx, q, k, v, mask, scale, precision="f32"):
        _lib.require_cuda(q, k, v)
        q, k, v = q.float().contiguous(), k.float().contiguous(), v.float().contiguous()
        BH, Lq, D = q.shape
        Lk = k.shape[1]
        out = torch.empty_like(q)
        lse = torch.empty(BH, Lq, dtype=torch.float32, device=q.device)
        m8, per_bh = None, 0
        if mask is not None:
            m8 = mask.to(torch.uint8).contiguous()
            per_bh = 1 if m8.dim() == 3 else 0
        with _lib.on_device(q):
            if precision in ("bf16", "fp16"):
                ws = _workspace(BH, Lq, Lk, D, q.device)
                _lib.call("gcn_attention_fwd_" + ("bf16" if precision == "bf16" else "f16"), _lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(m8), per_bh, BH, Lq,
                          Lk, D, float(scale), _lib.ptr(out), _lib.ptr(lse), _lib.ptr(ws), _lib.stream_of(q),
                          tag="attention_fwd_%s[BH=%d,Lq=%d,Lk=%d,D=%d]" % (precision, BH, Lq, Lk, D))
            else:
                _lib.call("gcn_attention_fwd", _lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(m8), per_bh, BH, Lq, Lk,
                          D, float(scale), _lib.ptr(out), _lib.ptr(lse), _lib.stream_of(q),
                          tag="attention_fwd[BH=%d,Lq=%d,Lk=%d,D=%d]" % (BH, Lq, Lk, D))
        ctx.save_for_backward(q, k, v, out, lse, m8 if m8 is not None else torch.empty(0, device=q.device))
        ctx.scale = scale
        ctx.precision = precision
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse, m8 = ctx.saved_tensors
        if ctx.precision in ("bf16", "fp16"):
            BH, Lq, D = q.shape
            Lk = k.shape[1]
            dout = dout.float().contiguous()
            dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
            mm = m8 if m8.numel() else None
            with _lib.on_device(q):
                ws = _workspace(BH, Lq, Lk, D, q.device)
                _lib.call("gcn_attention_bwd_" + ("bf16" if ctx.precision == "bf16" else "f16"), _lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(out), _lib.ptr(dout),
                          _lib.ptr(lse), _lib.ptr(mm), 1 if (mm is not None and mm.dim() == 3) else 0, BH, Lq, Lk, D,
                          float(ctx.scale), _lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dv), _lib.ptr(ws), _lib.stream_of(q),
                          tag="attention_bwd_%s[BH=%d,Lq=%d,Lk=%d,D=%d]" % (ctx.precision, BH, Lq, Lk, D))
            return dq, dk, dv, None, None, None
        s = torch.bmm(q, k.transpose(1, 2)) * ctx.scale
        if m8.numel():
            s = s.masked_fill(m8.bool() if m8.dim() == 3 else m8.bool().unsqueeze(0), float("-inf"))
        p = torch.exp(s - lse.unsqueeze(-1))                      # recomputed probabilities
        dv = torch.bmm(p.transpose(1, 2), dout)
        dp = torch.bmm(dout, v.transpose(1, 2))
        delta = (dout * out).sum(-1, keepdim=True)
        ds = p * (dp - delta) * ctx.scale
        return torch.bmm(ds, k), torch.bmm(ds.transpose(1, 2), q), dv, None, None, None


def sdpa(q, k, v, mask=None, scale=None, precision="f32"):
    """precision "f32": exact kernel (parity path, head dim 8..64); "bf16" / "fp16": matrix-core flash kernels
    (head dim 32/64) on bf16 or IEEE-half operands (BASELINE config 5 names fp16)."""
    if scale is None:
        scale = q.shape[-1] ** -0.5
    return SDPAFunction.apply(q, k, v, mask, scale, precision)
