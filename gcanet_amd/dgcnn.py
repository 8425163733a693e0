"""Host-side mirror of the reference model's hot path
(models/dgcnn-hais-concat-direct-4.py, "M4"): same function names, argument meaning and
tensor layouts, with the heavy lifting done by libgcanet_hip.so.
"""
import numpy as np
import torch

from . import _lib


def _zeroed_like(shape, dtype, device):
    from .layers import zeroed_like
    return zeroed_like(shape, dtype, device)


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
    tile_ws = None
    # 3-D clouds: Morton-tiled kernel with box pruning (identical results).  Not used for the normal metric: its
    # factor (3 - 2 n_i.n_j) in [1,5] inflates the k-th key relative to the Euclidean bound, and with incoherent
    # normals (the synthetic benchmark clouds) more than half of the tiles survive -- slower than the full scan.
    # 3-D clouds (xyz, or xyz + normal), 1024 <= N <= 16384, k <= 128: threshold + filter + re-rank in exact arithmetic
    # (csrc/knn_normal.hip)
    if ((metric == 1 and C == 6) or (metric == 0 and C == 3)) and _lib.lib().gcn_knn_normal_supported(B, N, k2):
        # scratch per call: the caching allocator makes that free, and inside a HIP-graph capture the buffer lands in the
        # graph's private pool and stays valid for every replay (a process-global cache handed replays a stale address
        # once a later call had outgrown and replaced the buffer)
        tile_ws = torch.empty(_lib.lib().gcn_knn_tiles_ws_bytes(B, C, N), dtype=torch.uint8, device=x.device)
    elif k2 <= 64 and N >= 512 and metric == 0 and C == 3:
        tile_ws = torch.empty(_lib.lib().gcn_knn_tiles_ws_bytes(B, C, N), dtype=torch.uint8, device=x.device)
    with _lib.on_device(x):
        _lib.call("gcn_knn_model", _lib.ptr(x), B, C, N, k1, k2, metric, _lib.ptr(idx), None, _lib.ptr(xx),
                  _lib.ptr(tile_ws), _lib.stream_of(x), tag="knn_model[B=%d,C=%d,N=%d,k=%d]" % (B, C, N, k2))
    return idx


def knn_feature_pm(x_pm, k1, k2, stats=None):
    """`knn` on POINT-major features x_pm (B,N,C) f32, C in {32,64,128}: bf16 matrix-core prefilter + exact f32
    re-rank (csrc/knn_filter.hip) -- bit-identical indices to `knn`.  Returns None when the shape is not served.
    stats: an optional dict that receives `flagged` (queries that went to the exhaustive stage) and `candidates`
    (prefilter survivors over all queries) of this call -- diagnostics, synchronises."""
    B, N, C = x_pm.shape
    lib = _lib.lib()
    if not (x_pm.is_cuda and lib.gcn_knn_feature_supported(B, N, C, k2)):
        return None
    x_pm = x_pm.float().contiguous()
    step = k2 // k1
    kout = len(range(0, k2, step))
    ws = torch.empty(lib.gcn_knn_feature_ws_bytes(B, N, C), dtype=torch.uint8, device=x_pm.device)   # per call, see _knn_model
    idx = torch.empty(B, N, kout, dtype=torch.int64, device=x_pm.device)
    with _lib.on_device(x_pm):
        _lib.call("gcn_knn_feature", _lib.ptr(x_pm), B, N, C, k1, k2, _lib.ptr(idx), _lib.ptr(ws),
                  _lib.stream_of(x_pm), tag="knn_model[B=%d,C=%d,N=%d,k=%d]" % (B, C, N, k2))
        if stats is not None:
            import ctypes
            fl, ca = ctypes.c_long(0), ctypes.c_long(0)
            _lib.call("gcn_knn_feature_stats", _lib.ptr(ws), B, N, C, ctypes.addressof(fl), ctypes.addressof(ca), _lib.stream_of(x_pm))
            stats.update(flagged=fl.value, candidates=ca.value)
    return idx


def knn(x, k1, k2, x_pm=None):
    """M4:30-47: x (B,C,N) -> idx (B,N,k1) int64: the k2 nearest in feature space (self included),
    every (k2//k1)-th kept.  One fused launch sequence per call for the whole batch; the reference loops over the
    batch in Python and materialises N x N per cloud.  Ties -> lowest index.
    x_pm: the same features point-major (B,N,C), when the caller already has them."""
    with torch.no_grad():
        if x.is_cuda and x.dim() == 3 and _lib.lib().gcn_knn_feature_supported(x.shape[0], x.shape[2], x.shape[1], k2):
            idx = knn_feature_pm(x.transpose(1, 2) if x_pm is None else x_pm, k1, k2)
            if idx is not None:
                return idx
        return _knn_model(x, k1, k2, 0)


def knn_points_normals(x, k1, k2):
    """M4:50-90: metric |p_i-p_j|^2 * (1 + (2 - 2 n_i.n_j)) on x = [xyz; normal] (B,6,N)."""
    with torch.no_grad():
        return _knn_model(x, k1, k2, 1)


# ------------------------------------------------------------------------------------------
# Fused EdgeConv block: get_graph_feature -> Conv2d 1x1 -> GroupNorm -> LeakyReLU -> max_k
# ------------------------------------------------------------------------------------------
def _run(name, like, *args, tag=None):
    with _lib.on_device(like):
        _lib.call(name, *args, _lib.stream_of(like), tag=tag)


def _edgeconv_dtype(dtype, C, Cout, groups):
    """The kernel that serves this layer: the bf16 matrix-core kernel needs Cout in {64,128}, (Cout/groups) % 32 == 0 and
    at most 128 (padded) input channels -- 256 with Cout == 128 (BASELINE configs[4]); any other width (M4:493-505 takes
    arbitrary channels) runs on the exact f32 kernel -- same op, same signature, no error."""
    Cp = _lib.lib().gcn_edgeconv_padded_channels(C)
    if dtype in ("bf16", "f16") and not (Cout in (64, 128) and (Cout // groups) % 32 == 0 and Cout % groups == 0
                                         and (Cp <= 128 or (Cp == 256 and Cout == 128))):
        return "f32"
    if dtype == "f16" and Cp < 64:          # the IEEE-half instantiations start at 33 input channels
        return "bf16"
    return dtype


_T16 = {"bf16": (torch.bfloat16, 0, 1), "f16": (torch.float16, 1, 2)}      # torch type, `half` flag, gcn_edgeconv_fwd dtype


def _center_term(x_bf, wp, rows, C, Cout, k):
    """q (rows, Cout) f32 = x . (W2 - W1)^T, the per-point half of the EdgeConv contraction (csrc/edgeconv_fwd.hip);
    None for k > 128, where the kernel contracts full [x_j ; x_i] rows."""
    if k > 128:
        if x_bf.dtype == torch.float16:
            raise ValueError("EdgeConv on IEEE-half operands serves k <= 128 (got k=%d)" % k)
        return None
    q = torch.empty(rows, Cout, dtype=torch.float32, device=x_bf.device)
    _run("gcn_edgeconv_center_f16" if x_bf.dtype == torch.float16 else "gcn_edgeconv_center", x_bf, _lib.ptr(x_bf),
         _lib.ptr(wp), rows, C, Cout, _lib.ptr(q))
    return q


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
    gsum = _zeroed_like((B, groups, 2), torch.float64, dev)
    x_pm = torch.empty(B, N, C, **f32)
    dtype = _edgeconv_dtype(dtype, C, Cout, groups)
    if dtype in _T16:
        t16, half, code = _T16[dtype]
        Cp = _lib.lib().gcn_edgeconv_padded_channels(C)
        x_bf = torch.empty(B, N, Cp, dtype=t16, device=dev)
        wp = torch.empty(Cout, 2 * Cp, dtype=t16, device=dev)
        _run("gcn_edgeconv_pack_x16", x, _lib.ptr(x), B, C, N, _lib.ptr(x_bf), _lib.ptr(x_pm), half)
        _run("gcn_edgeconv_pack_w16", x, _lib.ptr(w), Cout, C, _lib.ptr(wp), half)
        q = _center_term(x_bf, wp, B * N, C, Cout, k)
        _run("gcn_edgeconv_fwd", x, _lib.ptr(x_bf), _lib.ptr(wp), _lib.ptr(idx), code, B, N, N, C, k, Cout, groups,
             _lib.ptr(q), _lib.ptr(ymax), _lib.ptr(ymin), _lib.ptr(amax), _lib.ptr(amin), _lib.ptr(gsum), None,
             tag="edgeconv_fwd[B=%d,N=%d,k=%d,C=%d,Cout=%d]" % (B, N, k, C, Cout))
    elif dtype == "f32":
        _run("gcn_edgeconv_pack_x", x, _lib.ptr(x), B, C, N, None, _lib.ptr(x_pm))
        _run("gcn_edgeconv_fwd", x, _lib.ptr(x_pm), _lib.ptr(w), _lib.ptr(idx), 0, B, N, N, C, k, Cout, groups,
             None, _lib.ptr(ymax), _lib.ptr(ymin), _lib.ptr(amax), _lib.ptr(amin), _lib.ptr(gsum), None)
    else:
        raise ValueError("dtype must be 'bf16', 'f16' or 'f32'")
    out = torch.empty(B, Cout, N, **f32)
    mean_rstd = torch.empty(B, groups, 2, **f32)
    ga, be = gamma.float().contiguous(), beta.float().contiguous()
    _run("gcn_edgeconv_finish", x, _lib.ptr(ymax), _lib.ptr(ymin), _lib.ptr(gsum), _lib.ptr(ga), _lib.ptr(be),
         B, N, k, Cout, groups, float(eps), float(slope), _lib.ptr(out), None, _lib.ptr(mean_rstd), None, 0)
    # gsum may live in the step's pre-zeroed arena (re-used next step): a tensor handed to the caller must not
    return dict(out=out, ymax=ymax, ymin=ymin, amax=amax, amin=amin, gsum=gsum.clone(), mean_rstd=mean_rstd, x_pm=x_pm)


class EdgeConvFunction(torch.autograd.Function):
    """Differentiable fused EdgeConv block.

    forward : csrc/edgeconv.hip (one grouped MFMA contraction + finish kernel).
    backward: exact gradient of  max_k LeakyReLU(GroupNorm(Conv1x1([x_j - x_i ; x_i])))  in closed
    form.  The upstream gradient reaches the conv output y through (a) ONE selected neighbour per
    (point, channel) and (b) the GroupNorm statistics, whose contribution is affine in y:
        dy[n,j,c] = coef[n,c]*[j == j*(n,c)] + A_c + B_c * y[n,j,c].
    Because y = W1.x_j + (W2-W1).x_i is linear, the affine part collapses onto three graph
    aggregations of x (neighbour sum s = Adj.x, reverse sum r = Adj^T.x, in-degree) and a few
    (N x C x C) GEMMs -- the (B,Cout,N,k) gradient tensor the reference's autograd materialises
    (2 GB at B=8,N=8192,k=64,C=128) never exists.
    """

    @staticmethod
    def forward(ctx, x, idx, weight, gamma, beta, groups, dtype, eps, slope):
        need = x.requires_grad or weight.requires_grad or gamma.requires_grad or beta.requires_grad
        r = edgeconv_forward_raw(x, idx, weight, gamma, beta, groups, dtype, eps, slope, need_arg=need)
        if need:
            ctx.save_for_backward(r["x_pm"], idx, weight, gamma, beta, r["ymax"], r["ymin"], r["amax"], r["amin"],
                                  r["mean_rstd"])
            ctx.cfg = (groups, slope)
        ctx.mark_non_differentiable(idx)
        return r["out"]

    @staticmethod
    def backward(ctx, dout):
        dx_pm, dW, dgamma, dbeta = _edgeconv_backward(ctx.saved_tensors, ctx.cfg, dout, pm=False)
        return dx_pm.permute(0, 2, 1).contiguous(), None, dW, dgamma, dbeta, None, None, None, None


def _ts2(a, b):
    from .layers import tall_skinny_tn
    return tall_skinny_tn(a, b)


def _edgeconv_backward(saved, cfg, dout, pm, need_dx=True):
    """Closed-form EdgeConv backward; dout (B,Cout,N) (pm=False) or (B,N,Cout) (pm=True) -> dx (B,N,C), dW, dgamma, dbeta.
    need_dx=False (first layer: the cloud itself carries no gradient) skips the transposed aggregation r = Adj^T.x and
    the GEMMs that only feed dx."""
    if True:
        x, idx, W, gamma, beta, ymax, ymin, amax, amin, mean_rstd = saved
        G, slope = cfg
        B, N, C = x.shape
        k = idx.shape[2]
        Cout = W.shape[0]
        W1, Wd = W[:, :C], W[:, C:] - W[:, :C]
        Mg = float((Cout // G) * N * k)
        dpm = dout if pm else dout.permute(0, 2, 1)
        _, coef, Ac, Bc, dgamma, dbeta, Dsp = _route_backward_fused(dpm, gamma, beta, ymax, ymin, amax, amin, mean_rstd,
                                                                    G, slope, Mg, idx=idx, want_dsp=True)
        # graph aggregations: s = Adj.x (gather), r = Adj^T.x and the in-degree (destination-partitioned LDS scatter)
        s = torch.empty_like(x)
        r = torch.empty_like(x) if need_dx else None
        indeg = torch.empty(B, N, dtype=torch.float32, device=x.device)
        _run("gcn_neighbor_sum", x, _lib.ptr(x), _lib.ptr(idx), B, N, C, k, _lib.ptr(s))
        ws4 = torch.empty(_lib.lib().gcn_reverse_sum_ws_bytes(B, N, C, k), dtype=torch.uint8, device=x.device)
        _run("gcn_reverse_sum", x, _lib.ptr(x), _lib.ptr(idx), B, N, C, k, _lib.ptr(r), _lib.ptr(indeg), _lib.ptr(ws4))
        SW, XW = s @ W1.t(), x @ Wd.t()                                  # (B,N,Cout) each
        if need_dx:
            P1, RW = x @ W1.t(), r @ Wd.t()
            D1, D2 = torch.empty_like(coef), torch.empty_like(coef)
            _run("gcn_edge_combine", x, _lib.ptr(coef), _lib.ptr(Dsp), _lib.ptr(indeg), _lib.ptr(Ac), _lib.ptr(Bc),
                 _lib.ptr(P1), _lib.ptr(SW), _lib.ptr(XW), _lib.ptr(RW), B, N, k, Cout, _lib.ptr(D1), _lib.ptr(D2))
            dx_pm = torch.baddbmm(D1 @ W1, D2, Wd.unsqueeze(0).expand(B, -1, -1))   # D1.W1 + D2.Wd  (B,N,C), no separate add
        else:
            D2 = torch.empty_like(coef)                               # one kernel instead of six elementwise launches
            _run("gcn_edge_combine", x, _lib.ptr(coef), None, None, _lib.ptr(Ac), _lib.ptr(Bc), None, _lib.ptr(SW), _lib.ptr(XW),
                 None, B, N, k, Cout, None, _lib.ptr(D2))
            dx_pm = None
        # weight gradients
        if (C <= 16 or C == 64) and Cout in (64, 128):                 # all row reductions in one MFMA pass
            dW = torch.empty(Cout, 2 * C, dtype=torch.float32, device=x.device)
            wsf = torch.empty(_lib.lib().gcn_edge_wgrad_ws_floats(B, C, Cout), dtype=torch.float32, device=x.device)   # fully overwritten
            _run("gcn_edge_wgrad", x, _lib.ptr(x), _lib.ptr(s), _lib.ptr(Dsp), _lib.ptr(D2), _lib.ptr(indeg),
                 _lib.ptr(W.contiguous()), _lib.ptr(Ac), _lib.ptr(Bc), B, N, C, Cout, _lib.ptr(dW), _lib.ptr(wsf))
            return dx_pm, dW, dgamma, dbeta
        G11 = _tall_skinny_tn(x * indeg.unsqueeze(2), x)               # X^T diag(indeg) X   (B,C,C)
        G21 = _tall_skinny_tn(x, s)                                    # X^T S
        ssum = s.sum(1)                                                # (B,C)
        xf2 = x.reshape(B * N, C)
        dW1 = _ts2(Dsp.reshape(B * N, Cout), xf2) + torch.einsum("bo,bc->oc", Ac, ssum) \
            + torch.einsum("bo,boc->oc", Bc, torch.einsum("oc,bcd->bod", W1, G11) + torch.einsum("oc,bcd->bod", Wd, G21))
        dWd = _ts2(D2.reshape(B * N, Cout), xf2)
        dW = torch.cat([dW1 - dWd, dWd], 1)
        return dx_pm, dW, dgamma, dbeta


def edge_conv(x, idx, weight, gamma, beta, groups=2, dtype="bf16", eps=1e-5, slope=0.2):
    """out (B,Cout,N) = max_k LeakyReLU(GroupNorm(Conv2d_1x1(get_graph_feature(x, idx))))  (M4:493-505).
    x (B,C,N) f32, idx (B,N,k) int64, weight (Cout,2C) or the Conv2d's (Cout,2C,1,1)."""
    if weight.dim() == 4:
        weight = weight.flatten(1)                      # (Cout,Cin,1,1) -> (Cout,Cin): a view, no select/backward-fill kernels
    return EdgeConvFunction.apply(x, idx, weight, gamma, beta, groups, dtype, eps, slope)


# ------------------------------------------------------------------------------------------
# Reference-compatible graph-feature functions (materialising; for drop-in users of M4:93-205).
# The fused path (edge_conv / grouped_block) never builds these tensors.
# ------------------------------------------------------------------------------------------
def _neighbours(x, idx):
    B, C, N = x.shape
    xt = x.transpose(2, 1).contiguous()
    nb = torch.gather(xt.unsqueeze(1).expand(-1, N, -1, -1), 2, idx.unsqueeze(-1).expand(-1, -1, -1, C))
    return nb, xt.unsqueeze(2)


def get_graph_feature(x, k1=20, k2=20, idx=None):
    """M4:93-124 -> (B,2C,N,k) = cat(x_j - x_i, x_i)."""
    if idx is None:
        idx = knn(x, k1, k2)
    nb, ctr = _neighbours(x, idx)
    ctr = ctr.expand_as(nb)
    return torch.cat((nb - ctr, ctr), dim=3).permute(0, 3, 1, 2)


def get_graph_feature_with_normals(x, k1=20, k2=20, idx=None):
    """M4:127-161."""
    if idx is None:
        idx = knn_points_normals(x, k1, k2)
    return get_graph_feature(x, k1, k2, idx)


def get_graph_feature_with_normals_g(x, k1=20, k2=20, idx=None):
    """M4:164-205 -> (B,7,N,k) = [clamp(n_i.n_j, +-0.99), n_j - n_i, n_i]."""
    if idx is None:
        idx = knn_points_normals(x, k1, k2)
    nb, ctr = _neighbours(x, idx)
    n_j, n_i = nb[..., 3:6], ctr[..., 3:6].expand(-1, -1, idx.shape[2], -1)
    angle = (n_i * n_j).sum(-1, keepdim=True).clamp(-0.99, 0.99)
    return torch.cat((angle, n_j - n_i, n_i), dim=3).permute(0, 3, 1, 2)


class GroupedBlockFunction(torch.autograd.Function):
    """max_k LeakyReLU(GroupNorm(Conv2d_1x1(ef))) for an ARBITRARY materialised edge feature
    ef (B,N,k,F) (e.g. the 7-channel normal feature, M4:575-577,691-693).  Forward reuses the fused
    MFMA kernel (rows are their own 'neighbours'); backward is the same closed form as EdgeConv."""

    @staticmethod
    def forward(ctx, ef, weight, gamma, beta, groups, eps, slope, dtype="bf16", pm_out=False):
        B, N, k, F = ef.shape
        Cout = weight.shape[0]
        dev = ef.device
        rows = ef.reshape(B, N * k, F).permute(0, 2, 1).contiguous()          # (B,F,N*k) "cloud" of edge rows
        ident = (torch.arange(N * k, device=dev, dtype=torch.int64).view(1, N, k)).expand(B, -1, -1).contiguous()
        w2 = torch.cat([weight, weight], 1)                                   # W' = [W | 0]
        f32 = dict(dtype=torch.float32, device=dev)
        ymax, ymin = torch.empty(B, N, Cout, **f32), torch.empty(B, N, Cout, **f32)
        amax = torch.empty(B, N, Cout, dtype=torch.uint8, device=dev)
        amin = torch.empty(B, N, Cout, dtype=torch.uint8, device=dev)
        gsum = _zeroed_like((B, groups, 2), torch.float64, dev)
        # N "points" whose k neighbours are rows n*k..n*k+k-1 of the edge-row "cloud"
        dtype = _edgeconv_dtype("bf16" if dtype == "f16" else dtype, F, Cout, groups)    # generic block: bf16 or exact
        if dtype == "bf16":
            Fp = _lib.lib().gcn_edgeconv_padded_channels(F)
            x_bf = torch.empty(B, N * k, Fp, dtype=torch.bfloat16, device=dev)
            wp = torch.empty(Cout, 2 * Fp, dtype=torch.bfloat16, device=dev)
            _run("gcn_edgeconv_pack_x", ef, _lib.ptr(rows), B, F, N * k, _lib.ptr(x_bf), None)
            _run("gcn_edgeconv_pack_w", ef, _lib.ptr(w2.contiguous()), Cout, F, _lib.ptr(wp))   # W' = [W | 0]
            _run("gcn_edgeconv_fwd", ef, _lib.ptr(x_bf), _lib.ptr(wp), _lib.ptr(ident), 1, B, N, N * k, F, k, Cout,
                 groups, None, _lib.ptr(ymax), _lib.ptr(ymin), _lib.ptr(amax), _lib.ptr(amin), _lib.ptr(gsum), None)
        else:   # exact path: W.(x_j - x_i) + W.x_i
            flat = ef.reshape(B, N * k, F)
            _run("gcn_edgeconv_fwd", ef, _lib.ptr(flat), _lib.ptr(w2.contiguous()), _lib.ptr(ident), 0, B, N, N * k, F,
                 k, Cout, groups, None, _lib.ptr(ymax), _lib.ptr(ymin), _lib.ptr(amax), _lib.ptr(amin), _lib.ptr(gsum), None)
        out_cm, out_pm, mean_rstd = _finish(ymax, ymin, gsum, gamma.float().contiguous(), beta.float().contiguous(),
                                            B, N, k, Cout, groups, eps, slope, not pm_out, pm_out)
        ctx.save_for_backward(ef, weight, gamma, beta, ymax, ymin, amax, amin, mean_rstd)
        ctx.cfg = (groups, slope, pm_out)
        return out_pm if pm_out else out_cm

    @staticmethod
    def backward(ctx, dout):
        ef, W, gamma, beta, ymax, ymin, amax, amin, mean_rstd = ctx.saved_tensors
        G, slope, pm_out = ctx.cfg
        B, N, k, F = ef.shape
        Cout = W.shape[0]
        dpm = dout if pm_out else dout.permute(0, 2, 1)
        jsel, coef, Ac, Bc, dgamma, dbeta, _ = _route_backward_fused(dpm, gamma.float().contiguous(), beta.float().contiguous(),
                                                                     ymax, ymin, amax, amin, mean_rstd, G, slope,
                                                                     float((Cout // G) * N * k), want_jsel=True)
        # sparse part: one selected edge row per (point, channel) -- never a (B,N,k,Cout) one-hot
        jx = jsel.unsqueeze(-1).expand(-1, -1, -1, F)                          # (B,N,Cout,F)
        dW = torch.einsum("bno,bnof->of", coef, torch.gather(ef, 2, jx))
        d_ef = None
        if ctx.needs_input_grad[0]:     # the normal-feature branch builds ef from the INPUT cloud: no gradient wanted
            contrib = coef.unsqueeze(-1) * W.view(1, 1, Cout, F)               # coef[n,c] * W[c,:]
            d_ef = torch.zeros_like(ef).scatter_add_(2, jx, contrib)           # (B,N,k,F)
            # dense part: dy = A + B*y, y = ef.W^T
            T = torch.einsum("of,bo,og->bfg", W, Bc, W)                        # (B,F,F)
            d_ef = d_ef + (Ac @ W).view(B, 1, 1, F) + torch.einsum("bnkf,bfg->bnkg", ef, T)
        rows = ef.reshape(B, N * k, F)
        gram = _tall_skinny_tn(rows, rows)                                     # (B,F,F), split-K over the N*k rows
        dW = dW + torch.einsum("bo,bf->of", Ac, rows.sum(1)) + torch.einsum("bo,og,bgf->of", Bc, W, gram)
        return d_ef, dW, dgamma, dbeta, None, None, None, None, None


class NormalEdgeBlockFunction(torch.autograd.Function):
    """The normal-feature EdgeConv block  max_k LeakyReLU(GroupNorm(Conv2d_1x1(get_graph_feature_with_normals_g(x))))
    (M4:164-205,575-577,691-693) on pts (B,N,6) point-major + idx (B,N,k), with the 7-channel edge feature rebuilt in
    registers (csrc/normaledge.hip) -- no (B,7,N,k) tensor.  pts is the input cloud: no gradient flows to it."""

    @staticmethod
    def forward(ctx, pts, idx, weight, gamma, beta, groups, eps, slope, pm_out):
        _lib.require_cuda(pts, idx)
        B, N, _ = pts.shape
        k = idx.shape[2]
        Cout = weight.shape[0]
        dev = pts.device
        pts, idx = pts.float().contiguous(), idx.contiguous()
        w = weight.float().reshape(Cout, 7).contiguous()
        f32 = dict(dtype=torch.float32, device=dev)
        # ROUTED forward: only the extreme GroupNorm+LeakyReLU will select is kept (ymin/amin not produced)
        ymax = torch.empty(B, N, Cout, **f32)
        amax = torch.empty(B, N, Cout, dtype=torch.uint8, device=dev)
        gsum = _zeroed_like((B, groups, 2), torch.float64, dev)
        ga, be = gamma.float().contiguous(), beta.float().contiguous()
        _run("gcn_normal_edge_fwd", pts, _lib.ptr(pts), _lib.ptr(idx), _lib.ptr(w), B, N, k, Cout, groups, _lib.ptr(ymax),
             None, _lib.ptr(amax), None, _lib.ptr(gsum), _lib.ptr(ga))
        out_cm, out_pm, mean_rstd = _finish(ymax, None, gsum, ga, be, B, N, k, Cout, groups, eps, slope, not pm_out, pm_out)
        ymin = amin = torch.empty(0, device=dev)
        ctx.save_for_backward(pts, idx, w, ga, be, ymax, ymin, amax, amin, mean_rstd)
        ctx.cfg = (groups, slope, pm_out, weight.shape)
        return out_pm if pm_out else out_cm

    @staticmethod
    def backward(ctx, dout):
        pts, idx, W, gamma, beta, ymax, ymin, amax, amin, mean_rstd = ctx.saved_tensors
        G, slope, pm_out, wshape = ctx.cfg
        B, N, k = idx.shape
        Cout = W.shape[0]
        dpm = dout if pm_out else dout.permute(0, 2, 1)
        jsel, coef, Ac, Bc, dgamma, dbeta, _ = _route_backward_fused(dpm, gamma, beta, ymax, ymin, amax, amin, mean_rstd, G,
                                                                     slope, float((Cout // G) * N * k), want_jsel=True)
        f32 = dict(dtype=torch.float32, device=pts.device)
        raw = _zeroed_like((B * Cout * 7 + B * 7 + B * 49,), torch.float32, pts.device)    # adjacent accumulators: one zero fill in the library
        n1, n2 = B * Cout * 7, B * Cout * 7 + B * 7
        dWsp, esum, gram = raw[:n1].view(B, Cout, 7), raw[n1:n2].view(B, 7), raw[n2:].view(B, 7, 7)
        _run("gcn_normal_edge_bwd", pts, _lib.ptr(pts), _lib.ptr(idx), _lib.ptr(coef), _lib.ptr(jsel), B, N, k, Cout,
             _lib.ptr(dWsp), _lib.ptr(esum), _lib.ptr(gram))
        dW = dWsp.sum(0) + Ac.t() @ esum + torch.einsum("bo,og,bgf->of", Bc, W, gram)
        return None, None, dW.reshape(wshape), dgamma, dbeta, None, None, None, None


def normal_edge_block(pts, idx, weight, gamma, beta, groups=2, eps=1e-5, slope=0.2, pm_out=False):
    """pts (B,N,6) [xyz, normal] point-major, idx (B,N,k) -> (B,Cout,N) or (B,N,Cout) with pm_out; f32 arithmetic."""
    return NormalEdgeBlockFunction.apply(pts, idx, weight, gamma, beta, groups, eps, slope, pm_out)


def grouped_block(ef, weight, gamma, beta, groups=2, eps=1e-5, slope=0.2, dtype="bf16", pm_out=False):
    """ef (B,N,k,F) point-major edge features -> (B,Cout,N), or (B,N,Cout) with pm_out."""
    if weight.dim() == 4:
        weight = weight.flatten(1)                      # (Cout,Cin,1,1) -> (Cout,Cin): a view, no select/backward-fill kernels
    return GroupedBlockFunction.apply(ef.float().contiguous(), weight, gamma, beta, groups, eps, slope, dtype, pm_out)


def _route_backward_fused(dout_pm, gamma, beta, ymax, ymin, amax, amin, mean_rstd, G, slope, count_per_group,
                          idx=None, want_jsel=False, want_dsp=False):
    """Single-kernel version of _gn_route_backward (csrc/graphbwd.hip) for point-major dout.
    Returns (jsel|None, coef, Ac, Bc, dgamma, dbeta, dsp|None)."""
    B, N, Cout = ymax.shape
    dev = ymax.device
    k = idx.shape[2] if idx is not None else 1
    cpg = Cout // G
    dout_pm = dout_pm.float().contiguous()
    if not ((Cout <= 256 and 256 % Cout == 0) or Cout % 256 == 0):
        # widths the fused kernel's thread mapping does not cover (it wants Cout | 256 or 256 | Cout): the same
        # quantities in torch ops -- arbitrary channel counts stay usable, only slower
        ymin_t = ymin if (ymin is not None and ymin.numel()) else ymax
        amin_t = amin if (amin is not None and amin.numel()) else amax
        jsel, coef, Ac, Bc, dgamma, dbeta = _gn_route_backward(dout_pm, gamma, beta, ymax, ymin_t, amax, amin_t, mean_rstd,
                                                               G, slope, count_per_group, pm=True)
        dsp = None
        if want_dsp:
            msel = torch.gather(idx, 2, jsel)
            dsp = torch.zeros(B, N, Cout, dtype=torch.float32, device=dev).scatter_add_(1, msel, coef)
        return (jsel if want_jsel else None), coef.contiguous(), Ac.contiguous(), Bc.contiguous(), dgamma, dbeta, dsp
    coef = torch.empty(B, N, Cout, dtype=torch.float32, device=dev)
    jsel = torch.empty(B, N, Cout, dtype=torch.int64, device=dev) if want_jsel else None
    dsp = torch.empty(B, N, Cout, dtype=torch.float32, device=dev) if want_dsp else None
    from .layers import _acc_buffers
    dsp_ws = None
    if want_dsp and (B * G * 2 * 8 + 2 * Cout * 4) % 16 == 0:
        # scratch of the partitioned LDS scatter, right behind the accumulators: one zero fill inside gcn_route_bwd
        S, dgamma, dbeta, dsp_ws = _acc_buffers(B * G * 2, Cout, dev, tail_bytes=_lib.lib().gcn_route_bwd_ws_bytes(B, N, Cout))
    else:
        S, dgamma, dbeta = _acc_buffers(B * G * 2, Cout, dev)
    ymin = ymin if (ymin is not None and ymin.numel()) else None
    amin = amin if (amin is not None and amin.numel()) else None
    Ac = torch.empty(B, Cout, dtype=torch.float32, device=dev)
    Bc = torch.empty(B, Cout, dtype=torch.float32, device=dev)
    part_ws = torch.empty(_lib.lib().gcn_route_bwd_part_bytes(B, N, Cout, G), dtype=torch.uint8, device=dev)
    _run("gcn_route_bwd", ymax, _lib.ptr(dout_pm), _lib.ptr(ymax), _lib.ptr(ymin), _lib.ptr(amax), _lib.ptr(amin),
         _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(mean_rstd), _lib.ptr(idx), B, N, k, Cout, G, float(slope),
         _lib.ptr(coef), _lib.ptr(jsel), None, _lib.ptr(dsp), _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(S),
         float(count_per_group), _lib.ptr(Ac), _lib.ptr(Bc), _lib.ptr(dsp_ws), _lib.ptr(part_ws))
    return jsel, coef, Ac, Bc, dgamma, dbeta, dsp


def _gn_route_backward(dout, gamma, beta, ymax, ymin, amax, amin, mean_rstd, G, slope, count_per_group, pm=False):
    """Shared first half of the closed-form backward of  max_k LeakyReLU(GroupNorm(y)) :
    returns (jsel, coef, Ac, Bc, dgamma, dbeta) with dy[n,j,c] = coef[n,c]*[j==jsel[n,c]] + Ac[c] + Bc[c]*y[n,j,c]."""
    Cout = gamma.shape[0]
    cpg = Cout // G
    B, N = ymax.shape[:2]
    dpm = dout if pm else dout.permute(0, 2, 1)
    pos = (gamma >= 0).view(1, 1, Cout)
    ysel = torch.where(pos, ymax, ymin)
    jsel = torch.where(pos, amax, amin).long()
    mean = mean_rstd[:, :, 0].repeat_interleave(cpg, 1).unsqueeze(1)
    rstd = mean_rstd[:, :, 1].repeat_interleave(cpg, 1).unsqueeze(1)
    yhat = (ysel - mean) * rstd
    z = yhat * gamma + beta
    gz = dpm * torch.where(z > 0, torch.ones_like(z), torch.full_like(z, slope))
    dbeta, dgamma = gz.sum((0, 1)), (gz * yhat).sum((0, 1))
    t = gz * gamma
    S1 = t.view(B, N, G, cpg).sum((1, 3))
    S2 = (t * yhat).view(B, N, G, cpg).sum((1, 3))
    rs, mu = mean_rstd[:, :, 1], mean_rstd[:, :, 0]
    Bc = (-(rs * rs) * S2 / count_per_group).repeat_interleave(cpg, 1)
    Ac = (-(rs * S1) / count_per_group).repeat_interleave(cpg, 1) - Bc * mu.repeat_interleave(cpg, 1)
    return jsel, t * rstd, Ac, Bc, dgamma, dbeta


def _tall_skinny_tn(A, Bm, chunks=64):
    """A^T @ B for A (B,N,P), B (B,N,Q) with N >> P,Q: split the long reduction into `chunks` batched
    GEMMs (rocBLAS picks a 1-workgroup-per-tile kernel for the direct call: 1.7 ms at N=8192)."""
    Bsz, N, P = A.shape
    Q = Bm.shape[2]
    if N % chunks != 0 or N // chunks < 16:
        return A.transpose(1, 2) @ Bm
    part = A.view(Bsz * chunks, N // chunks, P).transpose(1, 2) @ Bm.view(Bsz * chunks, N // chunks, Q)
    return part.view(Bsz, chunks, P, Q).sum(1)


class KeyEdgeBlockFunction(torch.autograd.Function):
    """max_k LeakyReLU(GroupNorm(att[n,j] * (U[m_j] - V[n])))  -- the grouped block of the offset module
    (csrc/edgeconv.hip: keyedge_fwd_kernel).  Backward: closed form; the dense GroupNorm coupling is
    expressed through the (N x NK) incidence matrices A1 = sum_j att [m_j=m], A2 = sum_j att^2 [m_j=m]."""

    @staticmethod
    def forward(ctx, att, kidx, U, V, gamma, beta, groups, eps, slope, pm_out=False):
        B, N, k = att.shape
        NK, Cout = U.shape[1], U.shape[2]
        dev = att.device
        att, kidx, U, V = att.float().contiguous(), kidx.contiguous(), U.float().contiguous(), V.float().contiguous()
        f32 = dict(dtype=torch.float32, device=dev)
        # ROUTED forward: only the extreme GroupNorm+LeakyReLU will select is kept (ymin/amin not produced)
        ymax = torch.empty(B, N, Cout, **f32)
        amax = torch.empty(B, N, Cout, dtype=torch.uint8, device=dev)
        gsum = _zeroed_like((B, groups, 2), torch.float64, dev)
        ga, be = gamma.float().contiguous(), beta.float().contiguous()
        _run("gcn_keyedge_fwd", att, _lib.ptr(att), _lib.ptr(kidx), _lib.ptr(U), _lib.ptr(V), B, N, k, NK, Cout, groups,
             _lib.ptr(ymax), None, _lib.ptr(amax), None, _lib.ptr(gsum), _lib.ptr(ga),
             tag="keyedge_fwd[B=%d,N=%d,k=%d,NK=%d,Cout=%d]" % (B, N, k, NK, Cout))
        out_cm, out_pm, mean_rstd = _finish(ymax, None, gsum, ga, be, B, N, k, Cout, groups, eps, slope, not pm_out, pm_out)
        ymin = amin = torch.empty(0, device=dev)
        ctx.save_for_backward(att, kidx, U, V, ga, be, ymax, ymin, amax, amin, mean_rstd)
        ctx.cfg = (groups, slope, pm_out)
        return out_pm if pm_out else out_cm

    @staticmethod
    def backward(ctx, dout):
        att, kidx, U, V, gamma, beta, ymax, ymin, amax, amin, mean_rstd = ctx.saved_tensors
        G, slope, pm_out = ctx.cfg
        B, N, k = att.shape
        NK, Cout = U.shape[1], U.shape[2]
        Mg = float((Cout // G) * N * k)
        dpm = dout if pm_out else dout.permute(0, 2, 1)
        jsel, coef, Ac, Bc, dgamma, dbeta, _ = _route_backward_fused(dpm, gamma, beta, ymax, ymin, amax, amin, mean_rstd,
                                                                     G, slope, Mg, want_jsel=True)
        if k <= 64:
            X = V @ (U * Bc.unsqueeze(1)).transpose(1, 2)                   # (B,N,NK) = (V o Bc).U^T, scaling the NK rows
            datt, dV = torch.empty_like(att), torch.empty_like(V)
            A2 = torch.empty(B, N, NK, dtype=torch.float32, device=att.device)
            raw = _zeroed_like((U.numel() + B * 2 * NK,), torch.float32, att.device)   # adjacent: one zero fill
            dUsp, T12 = raw[:U.numel()].view_as(U), raw[U.numel():].view(B, 2, NK)
            _run("gcn_keyedge_bwd", att, _lib.ptr(att), _lib.ptr(kidx), _lib.ptr(U), _lib.ptr(V), _lib.ptr(coef.contiguous()),
                 _lib.ptr(jsel.contiguous()), _lib.ptr(Ac.contiguous()), _lib.ptr(Bc.contiguous()), _lib.ptr(X.contiguous()),
                 B, N, k, NK, Cout, _lib.ptr(datt), _lib.ptr(dV), _lib.ptr(A2), _lib.ptr(dUsp), _lib.ptr(T12))
            dU = dUsp + Ac.unsqueeze(1) * T12[:, 0].unsqueeze(-1) \
                + Bc.unsqueeze(1) * (U * T12[:, 1].unsqueeze(-1) - _tall_skinny_tn(A2, V))
            return datt, None, dU, dV, dgamma, dbeta, None, None, None, None
        # generic-k fallback in torch ops (incidence matrices A1 = sum_j att [m_j=m], A2 = sum_j att^2 [m_j=m])
        A_, B_ = Ac.unsqueeze(1), Bc.unsqueeze(1)                       # (B,1,Cout)
        att_sel = torch.gather(att, 2, jsel)                            # (B,N,Cout)
        m_sel = torch.gather(kidx, 2, jsel)
        d_sel = torch.gather(U, 1, m_sel) - V
        ca = coef * att_sel
        a1, a2 = att.sum(2), (att * att).sum(2)
        z_nk = torch.zeros(B, N, NK, dtype=torch.float32, device=att.device)
        A1 = z_nk.scatter_add(2, kidx, att)
        A2 = z_nk.scatter_add(2, kidx, att * att)
        sum_att_y = A2 @ U - V * a2.unsqueeze(-1)
        dV = -(ca + A_ * a1.unsqueeze(-1) + B_ * sum_att_y)
        dU = torch.zeros_like(U).scatter_add_(1, m_sel, ca) + A1.sum(1).unsqueeze(-1) * A_ \
            + B_ * (U * A2.sum(1).unsqueeze(-1) - _tall_skinny_tn(A2, V))
        gat = lambda tab: torch.gather(tab.unsqueeze(1).expand(-1, N, -1), 2, kidx)      # (B,NK) -> (B,N,k)
        datt = torch.zeros_like(att).scatter_add_(2, jsel, coef * d_sel)
        datt = datt + gat((U * A_).sum(-1)) - (V * A_).sum(-1, keepdim=True)
        VBU = (V * B_) @ U.transpose(1, 2)                                                  # (B,N,NK)
        datt = datt + att * (gat((U * U * B_).sum(-1)) - 2 * torch.gather(VBU, 2, kidx)
                             + (V * V * B_).sum(-1, keepdim=True))
        return datt, None, dU, dV, dgamma, dbeta, None, None, None, None


# ------------------------------------------------------------------------------------------
# nn.Modules mirroring M4's hot path (same attribute / parameter names, so a reference
# state_dict's matching keys load unchanged).
# ------------------------------------------------------------------------------------------
import torch.nn as nn  # noqa: E402
import torch.nn.functional as F  # noqa: E402


class _EdgeConvLayer(nn.Module):
    """Holds Conv2d(2C->Cout,1x1,no bias)+GroupNorm like nn.Sequential(conv, bn, LeakyReLU) of M4:467-480
    (parameter names `0.weight`; the GroupNorm is shared with encoder.bnX as in the reference)."""

    def __init__(self, cin2, cout, gn):
        super().__init__()
        self.add_module("0", nn.Conv2d(cin2, cout, kernel_size=1, bias=False))
        self.add_module("1", gn)
        self.add_module("2", nn.LeakyReLU(negative_slope=0.2))

    def fused(self, x, idx, dtype):
        conv, gn = self._modules["0"], self._modules["1"]
        return edge_conv(x, idx, conv.weight, gn.weight, gn.bias, gn.num_groups, dtype, gn.eps, 0.2)


class DGCNNEncoderGn(nn.Module):
    """M4:455-534.  forward(x (B,Cin,N)) -> (B,1280,N); neighbour lists of the three layers are kept
    in `self.last_idx` (layer 1's is reused by the normal-feature branch instead of being recomputed,
    M4:691)."""

    def __init__(self, mode=0, nn_nb=80, input_channels=3, dtype="bf16"):
        super().__init__()
        self.k = nn_nb
        self.mode = mode
        self.dtype = dtype
        self.bn1, self.bn2, self.bn3 = nn.GroupNorm(2, 64), nn.GroupNorm(2, 64), nn.GroupNorm(2, 128)
        self.bn4, self.bn5 = nn.GroupNorm(4, 256), nn.GroupNorm(8, 1024)
        self.conv1 = _EdgeConvLayer(input_channels * 2 if mode == 5 else input_channels, 64, self.bn1)
        self.conv2 = _EdgeConvLayer(64 * 2, 64, self.bn2)
        self.conv3 = _EdgeConvLayer(64 * 2, 128, self.bn3)
        self.mlp1 = nn.Conv1d(256, 1024, 1)
        self.bnmlp1 = nn.GroupNorm(8, 1024)
        self.last_idx = None
        self.keep_feats = False        # True: also keep the inputs of the two feature-space searches in `last_feats`
        self.last_feats = None
        self.direct_slices = True      # finish kernels write the bf16 slices of cat(x1,x2,x3) (False: torch.cat + conversion)

    def forward_pm(self, x_cm, x_pm=None, idxs=None):
        """Point-major core: x_cm (B,Cin,N) feeds the kNN, x_pm (B,N,Cin) the row gathers.
        Returns (x_features (B,N,256) -- f32, or the autocast type under torch.autocast --, x4 (B,1024)).  idxs: optional (idx1, idx2, idx3) neighbour lists to use
        instead of searching (parity tests isolate the feature math from near-tie flips this way)."""
        from .layers import conv1x1, group_norm_relu_max
        k = self.k
        if x_pm is None:
            x_pm = x_cm.transpose(1, 2).contiguous()
        if idxs is not None:
            idx1, idx2, idx3 = [i.contiguous() for i in idxs]
        else:
            idx1 = knn_points_normals(x_cm, k, k) if self.mode == 5 else knn(x_cm, k, k)
        # feature-space kNN reads the point-major activations directly where csrc/knn_filter.hip serves the shape;
        # otherwise the EdgeConv finish kernel also writes the channel-major copy the generic kNN kernels read
        B_, N_ = x_pm.shape[0], x_pm.shape[1]
        fast = bool(x_pm.is_cuda and _lib.lib().gcn_knn_feature_supported(B_, N_, 64, k))
        # under bf16 autocast both consumers of cat(x1,x2,x3) (mlp1 here, conv1's feature half in the caller) read it in
        # bf16: the three finish kernels write their bf16 column slice of that tensor directly (no cat, no conversion
        # pass), and the two input gradients meet in bf16 and come back as three slice conversions
        direct = bool(self.direct_slices and x_pm.is_cuda and torch.is_autocast_enabled()
                      and torch.get_autocast_dtype("cuda") == torch.bfloat16)
        xf16 = torch.empty(B_, N_, 256, dtype=torch.bfloat16, device=x_pm.device) if direct else None
        sl = (lambda a, b: xf16[:, :, a:b]) if direct else (lambda a, b: None)
        x1, x1_cm = edge_conv_pm(x_pm, idx1, self.conv1._modules["0"].weight, self.bn1, self.dtype, want_cm=not fast,
                                 bf_out=sl(0, 64))
        if idxs is None:
            idx2 = knn_feature_pm(x1.detach(), k, k) if fast else knn(x1_cm, k, k)
        x2, x2_cm = edge_conv_pm(x1, idx2, self.conv2._modules["0"].weight, self.bn2, self.dtype, want_cm=not fast,
                                 bf_out=sl(64, 128))
        if idxs is None:
            idx3 = knn_feature_pm(x2.detach(), k, k) if fast else knn(x2_cm, k, k)
        x3, _ = edge_conv_pm(x2, idx3, self.conv3._modules["0"].weight, self.bn3, self.dtype, want_cm=False,
                             bf_out=sl(128, 256))
        self.last_idx = (idx1, idx2, idx3)
        if self.keep_feats:
            self.last_feats = (x1.detach(), x2.detach())
        if direct:
            x_features = ConcatSlicesFunction.apply(xf16, x1, x2, x3)           # (B,N,256) bf16
        else:
            x_features = torch.cat((x1, x2, x3), dim=2)                        # (B,N,256)
            if x_features.is_cuda and torch.is_autocast_enabled():
                x_features = x_features.to(torch.get_autocast_dtype("cuda"))
        x4 = group_norm_relu_max(conv1x1(x_features, self.mlp1), self.bnmlp1)  # (B,1024); (B,N,1024) never written
        return x_features, x4

    def forward(self, x):
        """Reference signature: x (B,Cin,N) -> (B,1280,N)  (M4:492-534)."""
        B, _, N = x.shape
        xf, x4 = self.forward_pm(x)
        return torch.cat([x4.float().view(B, 1024, 1).expand(-1, -1, N), xf.float().transpose(1, 2)], 1)


def cos_dist(instance_feature, global_instance_feature):
    """M4:326-342."""
    if instance_feature.is_cuda and instance_feature.numel() > 0:
        from .layers import row_normalise                                       # one kernel each way instead of ~12
        a = row_normalise(instance_feature)
    else:
        a = instance_feature / instance_feature.norm(dim=-1, keepdim=True)
    b = global_instance_feature / global_instance_feature.norm(dim=-1, keepdim=True)
    return -(1 - torch.einsum("bnc,bkc->bnk", a, b))


class TopKRowsFunction(torch.autograd.Function):
    """torch.topk(x, k, dim=-1, largest=True) for short last dimensions (<= 128) through csrc/knn.hip:topk_rows_kernel;
    gradient = scatter of the value gradients, as torch's."""

    @staticmethod
    def forward(ctx, x, k):
        _lib.require_cuda(x)
        shp = x.shape
        NK = shp[-1]
        xc = x.contiguous()
        if xc.dtype not in (torch.float32, torch.bfloat16):
            xc = xc.float()
        R = xc.numel() // NK
        vals = torch.empty(R, k, dtype=torch.float32, device=x.device)
        idx = torch.empty(R, k, dtype=torch.int64, device=x.device)
        _run("gcn_topk_rows", xc, _lib.ptr(xc), 1 if xc.dtype == torch.bfloat16 else 0, R, NK, k, _lib.ptr(vals), _lib.ptr(idx))
        idx = idx.view(*shp[:-1], k)
        ctx.save_for_backward(idx)
        ctx.shape = shp
        ctx.mark_non_differentiable(idx)
        return vals.view(*shp[:-1], k).to(x.dtype), idx

    @staticmethod
    def backward(ctx, dvals, _):
        (idx,) = ctx.saved_tensors
        g = torch.zeros(ctx.shape, dtype=dvals.dtype, device=dvals.device)
        g.scatter_(-1, idx, dvals)
        return g, None


def topk_rows(x, k):
    if x.is_cuda and x.shape[-1] <= 128 and k <= 64:
        return TopKRowsFunction.apply(x, k)
    return torch.topk(x, k, dim=-1, largest=True)


class KPAM(nn.Module):
    """M4:351-373 (softmax over the k axis -- see oracle/ref_model.py:kpam)."""

    def __init__(self, C):
        super().__init__()
        self.dim = C
        self.conv1 = nn.Sequential(nn.Conv1d(C, C, kernel_size=1, bias=False), nn.ReLU(),
                                   nn.Conv1d(C, C, kernel_size=1, bias=False))

    def weights(self, attention_feature):
        # the two Conv1d(k -> k, 1x1) act along the last axis of the point-major (B,N,k) tensor: plain GEMMs with the same
        # parameters (no permutes, and no MIOpen convolution whose solver search lands on a naive kernel on a fresh box)
        from .layers import linear_pm
        a = linear_pm(torch.relu(linear_pm(attention_feature, self.conv1[0].weight.flatten(1))),
                      self.conv1[2].weight.flatten(1))
        return torch.softmax(a.float(), dim=2)                                 # (B,N,k)

    def forward(self, x, attention_feature):
        return self.weights(attention_feature).unsqueeze(-1) * x


_KEY_CACHE = {}


def key_point_indices(num_points, n_keys, device):
    """M4:403-406: np.random.seed(1234); shuffle(arange(N))[:n_keys] -- legacy NumPy RNG, cached."""
    key = (num_points, n_keys, str(device))
    if key not in _KEY_CACHE:
        st = np.random.get_state()
        l = np.arange(num_points)
        np.random.seed(1234)
        np.random.shuffle(l)
        np.random.set_state(st)
        _KEY_CACHE[key] = torch.from_numpy(l[:n_keys]).long().to(device)
    return _KEY_CACHE[key]


class OFFSET_PRED_MODULE(nn.Module):
    """M4:376-452.  The reference repeats key features to (B,N,120,128) (4 GB at B=8,N=8192) and runs
    topk twice; here every edge feature is att[n,j] * [f_key[m] ; p_key[m] - p_n], so the 131->128 conv
    factorises over the 120 key points: y[n,j,:] = att[n,j] * (U[m] - V[n]) with U = Wf.f_key + Wp.p_key
    (120 rows) and V = Wp.p_n -- identical values, k-fold fewer FLOPs, no (B,N,120,.) tensors."""

    def __init__(self, nn_nb=30, sampling_ratio=120):
        super().__init__()
        self.k = nn_nb
        self.sampling_ratio = sampling_ratio
        self.bn1 = nn.GroupNorm(2, 128)
        self.conv1 = nn.Sequential(nn.Conv2d(131, 128, kernel_size=1, bias=False), self.bn1,
                                   nn.LeakyReLU(negative_slope=0.2))
        self.attention = KPAM(nn_nb)
        self.mlp_offset = nn.Conv1d(256, 3, 1)

    def forward(self, points, feature, instance_feature, pm_out=False, topk_idx=None):
        """points (B,N,3), feature (B,N,128), instance_feature (B,N,64) -> offsets (B,3,N) [(B,N,3) if pm_out]."""
        B, N, _ = points.shape
        sub = key_point_indices(N, self.sampling_ratio, points.device)
        # index_select: its backward is one index_add (advanced indexing goes through a seven-kernel index_put chain)
        key_pts, key_feat, key_emb = (t.index_select(1, sub) for t in (points, feature, instance_feature))
        dist = cos_dist(instance_feature, key_emb)                             # (B,N,120)
        if topk_idx is None:
            topk_dist, topk_idx = topk_rows(dist, self.k)                      # once, not twice (M4:421-422)
        else:                                                                  # forced selection (parity tests)
            topk_dist = torch.gather(dist, 2, topk_idx)
        self.last_topk_idx = topk_idx
        att = self.attention.weights(topk_dist)                                # (B,N,k)
        W = self.conv1[0].weight.flatten(1)                                    # (128,131) view of the 1x1 kernel
        Wf, Wp = W[:, :128], W[:, 128:]
        U = key_feat @ Wf.t() + key_pts @ Wp.t()                               # (B,120,128)
        from .layers import linear_pm
        with torch.autocast("cuda", enabled=False):        # f32 (xyz precision); split-K weight gradient (65536-row reduction)
            V = linear_pm(points.float(), Wp.float())                          # (B,N,128)
        # fused: conv output att*(U[m]-V) -> GroupNorm -> LeakyReLU -> max over k, (B,N,k,128) never formed
        if pm_out:
            from .layers import conv1x1
            y = KeyEdgeBlockFunction.apply(att, topk_idx, U, V, self.bn1.weight, self.bn1.bias, self.bn1.num_groups,
                                           self.bn1.eps, 0.2, True)            # (B,N,128)
            # under autocast the 1x1 conv runs in the autocast type anyway: concatenate in that type (the f32
            # concatenation + its conversion moved 2.5x the bytes)
            dt = torch.get_autocast_dtype("cuda") if (y.is_cuda and torch.is_autocast_enabled()) else y.dtype
            return conv1x1(torch.cat([y.to(dt), feature.to(dt)], dim=2), self.mlp_offset)
        y = KeyEdgeBlockFunction.apply(att, topk_idx, U, V, self.bn1.weight, self.bn1.bias, self.bn1.num_groups,
                                       self.bn1.eps, 0.2)                      # (B,128,N)
        y = torch.cat([y, feature.permute(0, 2, 1).to(y.dtype)], dim=1)
        return self.mlp_offset(y)


class PrimitivesEmbeddingDGCNGn(nn.Module):
    """Hot-path part of M4:537-782 (`forward_train` up to and including `pt_offsets`): DGCNN encoder,
    per-point heads, normal-feature EdgeConv, embedding head, offset module.  Everything after
    (forward_grouping -> SoftGroup ops -> spconv tiny U-Net, M4:737-774) is data-dependent host code plus
    an un-vendored third-party sparse-conv library and stays outside this module."""

    def __init__(self, emb_size=64, num_primitives=10, mode=5, num_channels=6, nn_nb=80, dtype="bf16",
                 loss_class="r"):
        super().__init__()
        self.mode, self.nn_nb, self.dtype, self.loss_class = mode, nn_nb, dtype, loss_class
        self.encoder = DGCNNEncoderGn(mode=mode, nn_nb=nn_nb, input_channels=num_channels, dtype=dtype)
        self.offset_pred_block = OFFSET_PRED_MODULE(nn_nb=30, sampling_ratio=120)
        self.conv1, self.bn1 = nn.Conv1d(1024 + 256, 512, 1), nn.GroupNorm(8, 512)
        self.conv2, self.bn2 = nn.Conv1d(512, 256, 1), nn.GroupNorm(4, 256)
        self.conv3, self.bn3 = nn.Conv1d(262 if mode in (5, 3) else 259, 128, 1), nn.GroupNorm(4, 128)
        self.mlp_seg_prob1, self.mlp_seg_prob2 = nn.Conv1d(832, 256, 1), nn.Conv1d(256, emb_size, 1)
        self.bn_seg_prob1 = nn.GroupNorm(4, 256)
        self.bn_normal = nn.GroupNorm(2, 64)
        self.conv_normal = nn.Sequential(nn.Conv2d(7, 64, kernel_size=1, bias=False), self.bn_normal,
                                         nn.LeakyReLU(negative_slope=0.2))
        self.mlp_prim_prob1, self.mlp_prim_prob2 = nn.Conv1d(256, 256, 1), nn.Conv1d(256, num_primitives, 1)
        self.bn_prim_prob1 = nn.GroupNorm(4, 256)
        self.mlp_param_prob1, self.mlp_param_prob2 = nn.Conv1d(256, 256, 1), nn.Conv1d(256, 22, 1)
        self.bn_param_prob1 = nn.GroupNorm(4, 256)
        self.logsoftmax = nn.LogSoftmax(dim=1)
        self.keep_xf = False           # True: keep the encoder's (B,N,256) per-point features (with their graph) in `last_xf`
        self.last_xf = None

    @staticmethod
    def _unit(v):
        return v / (torch.norm(v, dim=-1, keepdim=True) + 1e-12)

    def forward(self, points, normals, idxs=None, topk_idx=None):
        """points, normals (B,N,3) -> dict(type_per_point (B,N,P), param_per_point (B,N,22), semantic_scores
        (B*N,P), pt_offsets (B*N,3), output_feats (B,N,emb)) -- the reference's shapes (M4:634-747).
        Activations are point-major (B,N,C) end to end; the reference's (B,C,N) Conv1d tensors are the same
        values transposed.  conv1 on cat[x4 repeated, x_features] (M4:510-511,644) is evaluated as
        W[:, :1024].x4 (once per cloud) + W[:, 1024:].x_features -- identical, 5x fewer FLOPs."""
        from .layers import (add_row_broadcast, cat_conv1x1_gn_relu, conv1x1, conv1x1_gn_relu, group_norm_relu, linear_pm,
                             param_normalise)
        B, N, _ = points.shape
        pts = torch.cat([points, normals], dim=-1).contiguous() if self.mode == 5 else points.contiguous()   # (B,N,6)
        pts_cm = pts.transpose(1, 2).contiguous()
        xf, x4 = self.encoder.forward_pm(pts_cm, pts, idxs=idxs)
        if self.keep_xf:
            self.last_xf = xf
        w1 = self.conv1.weight.flatten(1)
        h = add_row_broadcast(linear_pm(xf, w1[:, 1024:]), F.linear(x4, w1[:, :1024], self.conv1.bias))
        x = group_norm_relu(h, self.bn1)
        x_all = conv1x1_gn_relu(x, self.conv2, self.bn2)                                      # (B,N,256)
        x_type = conv1x1_gn_relu(x_all, self.mlp_prim_prob1, self.bn_prim_prob1)
        type_forgroup = conv1x1(x_type, self.mlp_prim_prob2)                                  # (B,N,P)
        type_per_point = F.log_softmax(type_forgroup.float(), dim=-1) if "r" in self.loss_class else type_forgroup
        x_para = conv1x1_gn_relu(x_all, self.mlp_param_prob1, self.bn_param_prob1)
        p = conv1x1(x_para, self.mlp_param_prob2)
        if p.shape[-1] == 22 and p.is_cuda:               # one kernel each way instead of the slice/norm/div/cat chain
            param_per_point = param_normalise(p)
        else:
            p = p.float()
            param_per_point = torch.cat([p[:, :, :4], self._unit(p[:, :, 4:7]), p[:, :, 7:8], self._unit(p[:, :, 8:11]),
                                         p[:, :, 11:15], self._unit(p[:, :, 15:18]), p[:, :, 18:22]], dim=2)
        # normal-feature EdgeConv: same input as encoder layer 1 -> same neighbour list (M4:691 recomputes it)
        idx1 = self.encoder.last_idx[0]
        if self.mode == 5 and not pts.requires_grad:      # fused: the (B,N,k,7) edge feature is never formed
            normal_feature = normal_edge_block(pts, idx1, self.conv_normal[0].weight, self.bn_normal.weight,
                                               self.bn_normal.bias, 2, self.bn_normal.eps, 0.2, pm_out=True)   # (B,N,64)
        else:
            bi = torch.arange(B, device=pts.device).view(B, 1, 1)
            n_i = pts[:, :, 3:6].unsqueeze(2)
            n_j = pts[bi, idx1][..., 3:6]                                                      # (B,N,k,3)
            angle = (n_i * n_j).sum(-1, keepdim=True).clamp(-0.99, 0.99)
            ef = torch.cat((angle, n_j - n_i, n_i.expand_as(n_j)), dim=3)                      # (B,N,k,7)
            normal_feature = grouped_block(ef, self.conv_normal[0].weight, self.bn_normal.weight, self.bn_normal.bias, 2,
                                           self.bn_normal.eps, 0.2, self.dtype, pm_out=True)   # (B,N,64)
        # cat -> (B,N,832) -> conv: the concatenation belongs to the layer, whose backward hands every part a contiguous
        # gradient (layers.py: CatLinearPMFunction)
        x = cat_conv1x1_gn_relu([x_all, x_type, x_para, normal_feature], self.mlp_seg_prob1, self.bn_seg_prob1)
        output_feats = conv1x1(x, self.mlp_seg_prob2).float()                                  # (B,N,emb)
        # (B,N,262) -> zero columns up to a multiple of 16: the GEMM kernel's k-step (the cat copies anyway)
        kin = x_all.shape[2] + pts.shape[2]
        parts = [x_all, pts]
        if pts.is_cuda and kin % 16:
            parts.append(torch.zeros(B, N, (kin + 15) // 16 * 16 - kin, dtype=x_all.dtype, device=pts.device))
        feat_plus = cat_conv1x1_gn_relu(parts, self.conv3, self.bn3)                           # (B,N,128); only x_all has a gradient
        semantic_scores = type_forgroup.reshape(-1, type_forgroup.shape[-1])
        feat_in = feat_plus if (feat_plus.is_cuda and torch.is_autocast_enabled()) else feat_plus.float()
        pt_offsets = self.offset_pred_block(pts[:, :, 0:3], feat_in, output_feats, pm_out=True, topk_idx=topk_idx)
        pt_offsets = pt_offsets.reshape(-1, 3)
        return dict(type_per_point=type_per_point, param_per_point=param_per_point,
                    semantic_scores=semantic_scores, pt_offsets=pt_offsets, output_feats=output_feats)


# ------------------------------------------------------------------------------------------
# Point-major (B,N,C) fast path: same math, no layout round trips.  Used by PrimitivesEmbeddingDGCNGn.
# ------------------------------------------------------------------------------------------
def _finish(ymax, ymin, gsum, gamma, beta, B, N, k, Cout, groups, eps, slope, want_cm, want_pm, bf_out=None):
    """bf_out: optional bf16 (B,N,Cout) column slice of a wider (B,N,W) buffer that also receives the result."""
    f32 = dict(dtype=torch.float32, device=ymax.device)
    out_cm = torch.empty(B, Cout, N, **f32) if want_cm else None
    out_pm = torch.empty(B, N, Cout, **f32) if want_pm else None
    mean_rstd = torch.empty(B, groups, 2, **f32)
    _run("gcn_edgeconv_finish", ymax, _lib.ptr(ymax), _lib.ptr(ymin), _lib.ptr(gsum), _lib.ptr(gamma), _lib.ptr(beta),
         B, N, k, Cout, groups, float(eps), float(slope), _lib.ptr(out_cm), _lib.ptr(out_pm), _lib.ptr(mean_rstd),
         _lib.ptr(bf_out), 0 if bf_out is None else bf_out.stride(1))
    return out_cm, out_pm, mean_rstd


class EdgeConvPMFunction(torch.autograd.Function):
    """EdgeConv on point-major x (B,N,C) f32 -> (out_pm (B,N,Cout), out_cm (B,Cout,N) or None).
    out_cm is a non-differentiable copy in the channel-major layout the kNN kernels read."""

    @staticmethod
    def forward(ctx, x, idx, weight, gamma, beta, groups, dtype, eps, slope, want_cm, bf_out=None):
        _lib.require_cuda(x, idx)
        B, N, C = x.shape
        k = idx.shape[2]
        Cout = weight.shape[0]
        dev = x.device
        x = x.float().contiguous()
        idx = idx.contiguous()
        w = weight.float().contiguous()
        ga, be = gamma.float().contiguous(), beta.float().contiguous()
        f32 = dict(dtype=torch.float32, device=dev)
        # ROUTED forward: only the extreme GroupNorm+LeakyReLU will select is kept (ymin/amin not produced)
        ymax = torch.empty(B, N, Cout, **f32)
        amax = torch.empty(B, N, Cout, dtype=torch.uint8, device=dev)
        ymin = amin = None
        gsum = _zeroed_like((B, groups, 2), torch.float64, dev)
        dtype = _edgeconv_dtype(dtype, C, Cout, groups)
        if dtype in _T16:
            t16, half, code = _T16[dtype]
            Cp = _lib.lib().gcn_edgeconv_padded_channels(C)
            x_bf = torch.empty(B, N, Cp, dtype=t16, device=dev)
            wp = torch.empty(Cout, 2 * Cp, dtype=t16, device=dev)
            _run("gcn_cast_pad16", x, _lib.ptr(x), B * N, C, _lib.ptr(x_bf), half)
            _run("gcn_edgeconv_pack_w16", x, _lib.ptr(w), Cout, C, _lib.ptr(wp), half)
            q = _center_term(x_bf, wp, B * N, C, Cout, k)
            _run("gcn_edgeconv_fwd", x, _lib.ptr(x_bf), _lib.ptr(wp), _lib.ptr(idx), code, B, N, N, C, k, Cout, groups,
                 _lib.ptr(q), _lib.ptr(ymax), None, _lib.ptr(amax), None, _lib.ptr(gsum), _lib.ptr(ga),
                 tag="edgeconv_fwd[B=%d,N=%d,k=%d,C=%d,Cout=%d]" % (B, N, k, C, Cout))
        else:
            _run("gcn_edgeconv_fwd", x, _lib.ptr(x), _lib.ptr(w), _lib.ptr(idx), 0, B, N, N, C, k, Cout, groups,
                 None, _lib.ptr(ymax), None, _lib.ptr(amax), None, _lib.ptr(gsum), _lib.ptr(ga))
        if bf_out is not None:
            assert bf_out.dtype == torch.bfloat16 and bf_out.shape == (B, N, Cout) and bf_out.stride(2) == 1 \
                and bf_out.stride(0) == N * bf_out.stride(1)
        out_cm, out_pm, mean_rstd = _finish(ymax, ymin, gsum, ga, be, B, N, k, Cout, groups, eps, slope, want_cm, True, bf_out)
        empty = torch.empty(0, device=dev)
        ctx.save_for_backward(x, idx, w, ga, be, ymax, empty, amax, empty, mean_rstd)
        ctx.cfg = (groups, slope)
        if out_cm is None:
            out_cm = torch.empty(0, device=dev)
        ctx.mark_non_differentiable(out_cm)
        return out_pm, out_cm

    @staticmethod
    def backward(ctx, dout_pm, _unused):
        dx_pm, dW, dgamma, dbeta = _edgeconv_backward(ctx.saved_tensors, ctx.cfg, dout_pm.contiguous(), pm=True,
                                                      need_dx=ctx.needs_input_grad[0])
        return dx_pm, None, dW, dgamma, dbeta, None, None, None, None, None, None


def edge_conv_pm(x_pm, idx, conv_weight, gn, dtype="bf16", want_cm=True, bf_out=None):
    """bf_out: a bf16 (B,N,Cout) column slice of the consumer's concatenated input; the finish kernel writes the result
    there as well (see ConcatSlicesFunction)."""
    w = conv_weight.flatten(1) if conv_weight.dim() == 4 else conv_weight
    return EdgeConvPMFunction.apply(x_pm, idx, w, gn.weight, gn.bias, gn.num_groups, dtype, gn.eps, 0.2, want_cm, bf_out)


class ConcatSlicesFunction(torch.autograd.Function):
    """torch.cat(parts, dim=2).to(buf.dtype) for f32 parts whose low-precision images ALREADY sit in the column slices of
    `buf` (written by the kernels that produced the parts): forward hands out `buf`, backward returns the gradient's
    column slices in the parts' type -- no concatenation pass, no conversion pass over the wide tensor either way."""

    @staticmethod
    def forward(ctx, buf, *parts):
        ctx.widths = [p.shape[2] for p in parts]
        ctx.dtypes = [p.dtype for p in parts]
        assert sum(ctx.widths) == buf.shape[2]
        return buf.view_as(buf)

    @staticmethod
    def backward(ctx, g):
        outs, c = [], 0
        for w, dt in zip(ctx.widths, ctx.dtypes):
            outs.append(g[:, :, c:c + w].to(dt).contiguous())
            c += w
        return (None, *outs)
