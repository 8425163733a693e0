"""Host-side mirror of the reference model's hot path
(models/dgcnn-hais-concat-direct-4.py, "M4"): same function names, argument meaning and
tensor layouts, with the heavy lifting done by libgcanet_hip.so.
"""
import numpy as np
import torch

from . import _lib


def _knn_model(x, k1, k2, metric):
    if x.dim() != 3:
        raise RuntimeError("knn: x must be (B, C, N)")
    _lib.require_cuda(x)
    x = x.float().contiguous()
    B, C, N = x.shape
    step = k2 // k1
    kout = len(range(0, k2, step))
    idx = torch.empty(B, N, kout, dtype=torch.int64, device=x.device)
    xx = torch.empty(B, N, dtype=torch.float32, device=x.device)
    with torch.cuda.device_of(x):
        _lib.call("gcn_knn_model", _lib.ptr(x), B, C, N, k1, k2, metric, _lib.ptr(idx), None, _lib.ptr(xx),
                  _lib.stream_of(x))
    return idx


def knn(x, k1, k2):
    """M4:30-47: x (B,C,N) -> idx (B,N,k1) int64: the k2 nearest in feature space (self included),
    every (k2//k1)-th kept.  One fused kernel per call for the whole batch; the reference loops over the
    batch in Python and materialises N x N per cloud.  Ties -> lowest index."""
    with torch.no_grad():
        return _knn_model(x, k1, k2, 0)


def knn_points_normals(x, k1, k2):
    """M4:50-90: metric |p_i-p_j|^2 * (1 + (2 - 2 n_i.n_j)) on x = [xyz; normal] (B,6,N)."""
    with torch.no_grad():
        return _knn_model(x, k1, k2, 1)


# ------------------------------------------------------------------------------------------
# Fused EdgeConv block: get_graph_feature -> Conv2d 1x1 -> GroupNorm -> LeakyReLU -> max_k
# ------------------------------------------------------------------------------------------
def _run(name, like, *args):
    with torch.cuda.device_of(like):
        _lib.call(name, *args, _lib.stream_of(like))


def edgeconv_forward_raw(x, idx, weight, gamma, beta, groups, dtype="bf16", eps=1e-5, slope=0.2, need_arg=False):
    """Low-level fused forward (csrc/edgeconv.hip).  x (B,C,N) f32, idx (B,N,k) int64,
    weight (Cout,2C) f32 [the Conv2d 1x1 weight of M4:463-465], gamma/beta (Cout).
    Returns dict(out (B,Cout,N), ymax, ymin, amax, amin, gsum, mean_rstd, x_pm)."""
    _lib.require_cuda(x, idx, weight, gamma, beta)
    B, C, N = x.shape
    k = idx.shape[2]
    Cout = weight.shape[0]
    assert weight.shape[1] == 2 * C and idx.dtype == torch.int64
    dev = x.device
    x = x.float().contiguous()
    idx = idx.contiguous()
    w = weight.float().contiguous()
    f32 = dict(dtype=torch.float32, device=dev)
    ymax, ymin = torch.empty(B, N, Cout, **f32), torch.empty(B, N, Cout, **f32)
    amax = amin = None
    if need_arg:
        amax = torch.empty(B, N, Cout, dtype=torch.uint8, device=dev)
        amin = torch.empty(B, N, Cout, dtype=torch.uint8, device=dev)
    gsum = torch.empty(B, groups, 2, dtype=torch.float64, device=dev)
    x_pm = torch.empty(B, N, C, **f32)
    if dtype == "bf16":
        Cp = _lib.lib().gcn_edgeconv_padded_channels(C)
        x_bf = torch.empty(B, N, Cp, dtype=torch.bfloat16, device=dev)
        wp = torch.empty(Cout, 2 * Cp, dtype=torch.bfloat16, device=dev)
        _run("gcn_edgeconv_pack_x", x, _lib.ptr(x), B, C, N, _lib.ptr(x_bf), _lib.ptr(x_pm))
        _run("gcn_edgeconv_pack_w", x, _lib.ptr(w), Cout, C, _lib.ptr(wp))
        _run("gcn_edgeconv_fwd", x, _lib.ptr(x_bf), _lib.ptr(wp), _lib.ptr(idx), 1, B, N, C, k, Cout, groups,
             _lib.ptr(ymax), _lib.ptr(ymin), _lib.ptr(amax), _lib.ptr(amin), _lib.ptr(gsum))
    elif dtype == "f32":
        _run("gcn_edgeconv_pack_x", x, _lib.ptr(x), B, C, N, None, _lib.ptr(x_pm))
        _run("gcn_edgeconv_fwd", x, _lib.ptr(x_pm), _lib.ptr(w), _lib.ptr(idx), 0, B, N, C, k, Cout, groups,
             _lib.ptr(ymax), _lib.ptr(ymin), _lib.ptr(amax), _lib.ptr(amin), _lib.ptr(gsum))
    else:
        raise ValueError("dtype must be 'bf16' or 'f32'")
    out = torch.empty(B, Cout, N, **f32)
    mean_rstd = torch.empty(B, groups, 2, **f32)
    ga, be = gamma.float().contiguous(), beta.float().contiguous()
    _run("gcn_edgeconv_finish", x, _lib.ptr(ymax), _lib.ptr(ymin), _lib.ptr(gsum), _lib.ptr(ga), _lib.ptr(be),
         B, N, k, Cout, groups, float(eps), float(slope), _lib.ptr(out), None, _lib.ptr(mean_rstd))
    return dict(out=out, ymax=ymax, ymin=ymin, amax=amax, amin=amin, gsum=gsum, mean_rstd=mean_rstd, x_pm=x_pm)
