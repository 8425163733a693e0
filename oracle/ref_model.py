"""CPU ORACLE (test infrastructure) -- plain PyTorch fp32 restatement of the pure-torch part
of the hot path: graph features, EdgeConv block, DGCNN encoder, offset module, attention
stacks.  Functional style (weights passed as dicts of tensors); every function cites the
reference lines it follows (M4 = models/dgcnn-hais-concat-direct-4.py).  Pinned against
tests/golden/*.npz, which were produced by running the reference's own code
(tests/golden/make_golden.py).  Never imported by gcanet_amd/.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import knn_model


# ----------------------------------------------------------------------------- kNN
def knn(x, k1, k2):
    """M4:30-47 (twin sppnet.py:14-31); ties -> lowest index (oracle convention)."""
    return torch.from_numpy(knn_model(x.detach().cpu().numpy(), k1, k2, 0))


def knn_points_normals(x, k1, k2):
    """M4:50-90."""
    return torch.from_numpy(knn_model(x.detach().cpu().numpy(), k1, k2, 1))


# ----------------------------------------------------------------------------- graph features
def _gather_neighbours(x, idx):
    """x (B,C,N), idx (B,N,k) -> neighbours (B,N,k,C), centre (B,N,1,C)  (M4:103-122)."""
    B, C, N = x.shape
    xt = x.transpose(2, 1).contiguous()                      # (B,N,C)
    flat = (idx + torch.arange(B).view(-1, 1, 1) * N).reshape(-1)
    nb = xt.reshape(B * N, C)[flat].view(B, N, idx.shape[2], C)
    return nb, xt.view(B, N, 1, C)


def get_graph_feature(x, k1=20, k2=20, idx=None):
    """M4:93-124: cat(x_j - x_i, x_i) -> (B,2C,N,k)."""
    if idx is None:
        idx = knn(x, k1, k2)
    nb, ctr = _gather_neighbours(x, idx)
    ctr = ctr.expand_as(nb)
    return torch.cat((nb - ctr, ctr), dim=3).permute(0, 3, 1, 2)


def get_graph_feature_with_normals(x, k1=20, k2=20, idx=None):
    """M4:127-161: same construction, neighbours from knn_points_normals."""
    if idx is None:
        idx = knn_points_normals(x, k1, k2)
    return get_graph_feature(x, k1, k2, idx)


def get_graph_feature_with_normals_g(x, k1=20, k2=20, idx=None):
    """M4:164-205: [clamp(n_i.n_j, +-0.99), n_j - n_i, n_i] -> (B,7,N,k)."""
    if idx is None:
        idx = knn_points_normals(x, k1, k2)
    nb, ctr = _gather_neighbours(x, idx)
    n_j, n_i = nb[..., 3:6], ctr[..., 3:6].expand(-1, -1, idx.shape[2], -1)
    angle = (n_i * n_j).sum(-1, keepdim=True).clamp(-0.99, 0.99)
    return torch.cat((angle, n_j - n_i, n_i), dim=3).permute(0, 3, 1, 2)


# ----------------------------------------------------------------------------- EdgeConv block
def edgeconv_block(x, idx, w, gamma, beta, groups, slope=0.2, eps=1e-5):
    """get_graph_feature -> Conv2d 1x1 (no bias) -> GroupNorm -> LeakyReLU -> max over k
    (M4:463-505).  x (B,C,N), idx (B,N,k), w (Cout,2C) -> (B,Cout,N)."""
    ef = get_graph_feature(x, idx=idx)
    y = F.conv2d(ef, w[:, :, None, None])
    y = F.leaky_relu(F.group_norm(y, groups, gamma, beta, eps), slope)
    return y.max(dim=-1)[0]


def grouped_block(ef, w, gamma, beta, groups, slope=0.2, eps=1e-5):
    """Same tail on a materialised edge tensor ef (B,Cin,N,k) (conv_normal M4:575-577, offset conv1 M4:391-393)."""
    y = F.conv2d(ef, w[:, :, None, None])
    y = F.leaky_relu(F.group_norm(y, groups, gamma, beta, eps), slope)
    return y.max(dim=-1)[0]


def dgcnn_encoder(x, sd, k, mode=5, idxs=None, prefix=""):
    """DGCNNEncoderGn.forward (M4:492-534; sppnet.py:180-225 returns (x4, x_features)).
    sd: state dict (conv{1,2,3}.0.weight, bn{1,2,3}.*, mlp1.*, bnmlp1.*); idxs: optional
    per-layer neighbour lists.  Returns (x4 (B,1024), x_features (B,256,N), [idx1,idx2,idx3])."""
    g = lambda n: torch.as_tensor(sd[prefix + n])
    used = []

    def layer(inp, i, first):
        if idxs is not None:
            idx = torch.as_tensor(idxs[i])
        elif first and mode == 5:
            idx = knn_points_normals(inp, k, k)
        else:
            idx = knn(inp, k, k)
        used.append(idx)
        w = g("conv%d.0.weight" % (i + 1))[:, :, 0, 0]
        return edgeconv_block(inp, idx, w, g("bn%d.weight" % (i + 1)), g("bn%d.bias" % (i + 1)), 2)

    x1 = layer(x, 0, True)
    x2 = layer(x1, 1, False)
    x3 = layer(x2, 2, False)
    xf = torch.cat((x1, x2, x3), dim=1)
    h = F.conv1d(xf, g("mlp1.weight"), g("mlp1.bias"))
    h = F.relu(F.group_norm(h, 8, g("bnmlp1.weight"), g("bnmlp1.bias")))
    return h.max(dim=2)[0], xf, used


# ----------------------------------------------------------------------------- offset module
def compute_batch_adjacency_matrix(pts, sigma=1.0):
    """M4:210-233 (dist_state=True): cdist, zero diagonal, GLOBAL min/max normalise, Gaussian, zero diagonal."""
    d = torch.cdist(pts, pts)
    d = d - torch.diag_embed(torch.diagonal(d, dim1=-2, dim2=-1))
    d = (d - d.min()) / (d.max() - d.min())
    a = torch.exp(-d ** 2 / (2 * sigma ** 2))
    return a - torch.diag_embed(torch.diagonal(a, dim1=-2, dim2=-1))


def cos_dist(a, b):
    """M4:326-342: -(1 - cos) between (B,N,C) and (B,K,C) -> (B,N,K)."""
    an = a / a.norm(dim=-1, keepdim=True)
    bn = b / b.norm(dim=-1, keepdim=True)
    return -(1 - torch.einsum("bnc,bkc->bnk", an, bn))


def key_point_indices(num_points, n_keys=120):
    """M4:403-406: legacy NumPy RNG, seed 1234, shuffle of arange(N), first n_keys."""
    l = np.arange(num_points)
    np.random.seed(1234)
    np.random.shuffle(l)
    return torch.from_numpy(l[:n_keys]).long()


def kpam(x, att, w1, w2):
    """KPAM.forward (M4:351-373).  att (B,N,k) is permuted to (B,k,N), run through
    Conv1d(k->k)-ReLU-Conv1d(k->k) (channels = the k axis), permuted BACK to (B,N,k) and only then
    soft-maxed with dim=2 -- i.e. over the k neighbours (the in-line comment 'b,c,n' at M4:363 is
    stale).  The weights scale x (B,N,k,F)."""
    a = att.permute(0, 2, 1)
    a = F.conv1d(F.relu(F.conv1d(a, w1[:, :, None])), w2[:, :, None]).permute(0, 2, 1)
    a = torch.softmax(a, dim=2).unsqueeze(-1)
    return a * x


def offset_pred_module(points, feature, emb, sd, nn_nb=30, n_keys=120, prefix="", topk_idx=None, info=None):
    """OFFSET_PRED_MODULE.forward (M4:398-452).  points (B,N,3), feature (B,N,128), emb (B,N,64)
    -> offsets (B,3,N).  Layout quirk kept: the conv runs on (B,131,k,N) and the max is over dim -2."""
    g = lambda n: torch.as_tensor(sd[prefix + n])
    B, N, _ = points.shape
    sub = key_point_indices(N, n_keys)
    key_pts, key_feat, key_emb = points[:, sub], feature[:, sub], emb[:, sub]
    dist = cos_dist(emb, key_emb)                                  # (B,N,120)
    if topk_idx is None:
        topk_dist, topk_idx = torch.topk(dist, nn_nb, dim=2, largest=True)
    else:       # a caller-chosen selection (parity tests feed the device's, after checking it against `dist`)
        topk_dist = torch.gather(dist, 2, topk_idx)
    if info is not None:
        info["cos_dist"] = dist
    bi = torch.arange(B).view(B, 1, 1)
    f = torch.cat([key_feat[bi, topk_idx], key_pts[bi, topk_idx] - points.unsqueeze(2)], 3)   # (B,N,k,131)
    f = kpam(f, topk_dist, g("attention.conv1.0.weight")[:, :, 0], g("attention.conv1.2.weight")[:, :, 0])
    y = F.conv2d(f.permute(0, 3, 2, 1), g("conv1.0.weight"))      # (B,128,k,N)
    y = F.leaky_relu(F.group_norm(y, 2, g("bn1.weight"), g("bn1.bias")), 0.2)
    y = y.max(dim=-2)[0]                                           # (B,128,N)
    y = torch.cat([y, feature.permute(0, 2, 1)], dim=1)            # (B,256,N)
    return F.conv1d(y, g("mlp_offset.weight"), g("mlp_offset.bias"))


# ----------------------------------------------------------------------------- attention stacks
def transformer(x, sd, depth, heads, prefix="", mask=None):
    """models/transformer.py:36-91: pre-norm MHSA (scale = dim**-0.5, NOT dim_head) + GELU FFN, residuals.
    mask (b, n-1) bool: padded with a leading True, outer product, masked scores filled with the FINITE -finfo.max
    (:57-62) -- a fully masked row therefore attends uniformly."""
    g = lambda n: torch.as_tensor(sd[prefix + n])
    dim = x.shape[-1]
    scale = dim ** -0.5
    for l in range(depth):
        p = "layers.%d." % l
        h = F.layer_norm(x, (dim,), g(p + "0.fn.norm.weight"), g(p + "0.fn.norm.bias"))
        qkv = F.linear(h, g(p + "0.fn.fn.to_qkv.weight"))
        b, n, _ = qkv.shape
        q, k, v = [t.view(b, n, heads, -1).transpose(1, 2) for t in qkv.chunk(3, dim=-1)]
        dots = q @ k.transpose(-1, -2) * scale
        if mask is not None:
            m = F.pad(mask.flatten(1), (1, 0), value=True)
            m = m[:, None, :] * m[:, :, None]
            dots = dots.masked_fill(~m.unsqueeze(1), -torch.finfo(dots.dtype).max)
        att = torch.softmax(dots, dim=-1)
        o = (att @ v).transpose(1, 2).reshape(b, n, -1)
        x = F.linear(o, g(p + "0.fn.fn.to_out.0.weight"), g(p + "0.fn.fn.to_out.0.bias")) + x
        h = F.layer_norm(x, (dim,), g(p + "1.fn.norm.weight"), g(p + "1.fn.norm.bias"))
        h = F.linear(F.gelu(F.linear(h, g(p + "1.fn.fn.net.0.weight"), g(p + "1.fn.fn.net.0.bias"))),
                     g(p + "1.fn.fn.net.3.weight"), g(p + "1.fn.fn.net.3.bias"))
        x = h + x
    return x


def _mha(q_in, k_in, v_in, w, b, ow, ob, nhead, attn_mask=None):
    """nn.MultiheadAttention(batch_first) forward, eval mode."""
    d = q_in.shape[-1]
    q = F.linear(q_in, w[:d], b[:d])
    k = F.linear(k_in, w[d:2 * d], b[d:2 * d])
    v = F.linear(v_in, w[2 * d:], b[2 * d:])
    B, L, _ = q.shape
    S = k.shape[1]
    hd = d // nhead
    q = q.view(B, L, nhead, hd).transpose(1, 2)
    k = k.view(B, S, nhead, hd).transpose(1, 2)
    v = v.view(B, S, nhead, hd).transpose(1, 2)
    s = q @ k.transpose(-1, -2) / math.sqrt(hd)
    if attn_mask is not None:
        s = s.masked_fill(attn_mask.view(1, 1, L, S), float("-inf"))
    o = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, L, d)
    return F.linear(o, ow, ob)


def query_decoder(x, batch_offsets, sd, num_layer, nhead, iter_pred=False, attn_mask=False, prefix=""):
    """models/query_decoder.py:104-239 (eval).  Quirks kept: in cross-attention the results of
    self.dropout(output) / self.norm(output) are DISCARDED (:40-42); pe only if present."""
    g = lambda n: torch.as_tensor(sd[prefix + n])
    has = lambda n: (prefix + n) in sd
    d = g("query.weight").shape[1]
    ln = lambda t, p: F.layer_norm(t, (d,), g(p + ".weight"), g(p + ".bias"))
    inst = F.relu(ln(F.linear(x, g("input_proj.0.weight"), g("input_proj.0.bias")), "input_proj.1"))
    maskf = F.linear(F.relu(F.linear(x, g("x_mask.0.weight"), g("x_mask.0.bias"))), g("x_mask.2.weight"), g("x_mask.2.bias"))
    B = len(batch_offsets) - 1
    query = g("query.weight").unsqueeze(0).repeat(B, 1, 1)
    pe = g("pe.weight").unsqueeze(0).repeat(B, 1, 1) if (iter_pred and has("pe.weight")) else None

    def head(q):
        qn = ln(q, "out_norm")
        mlp = lambda p: F.linear(F.relu(F.linear(qn, g(p + ".0.weight"), g(p + ".0.bias"))), g(p + ".2.weight"), g(p + ".2.bias"))
        masks, amasks = [], []
        for i in range(B):
            m = qn[i] @ maskf[batch_offsets[i]:batch_offsets[i + 1]].T
            if attn_mask:
                am = (m.sigmoid() < 0.5)
                am[torch.where(am.sum(-1) == am.shape[-1])] = False
                amasks.append(am)
            masks.append(m)
        return mlp("out_cls"), mlp("out_score"), masks, mlp("out_paras"), amasks

    outs = []
    amasks = []
    if iter_pred:
        o = head(query)
        outs.append(o)
        amasks = o[4]
    for l in range(num_layer):
        p = "cross_attn_layers.%d.attn." % l
        qpe = query if pe is None else query + pe
        new = []
        for i in range(B):
            kv = inst[batch_offsets[i]:batch_offsets[i + 1]].unsqueeze(0)
            am = amasks[i] if (iter_pred and amasks) else None
            o = _mha(qpe[i:i + 1], kv, kv, g(p + "in_proj_weight"), g(p + "in_proj_bias"),
                     g(p + "out_proj.weight"), g(p + "out_proj.bias"), nhead, am)
            new.append(o + qpe[i])
        query = torch.cat(new, 0)
        p = "self_attn_layers.%d." % l
        qk = query if pe is None else query + pe
        o = _mha(qk, qk, query, g(p + "attn.in_proj_weight"), g(p + "attn.in_proj_bias"),
                 g(p + "attn.out_proj.weight"), g(p + "attn.out_proj.bias"), nhead)
        query = ln(o + query, p + "norm")
        p = "ffn_layers.%d." % l
        o = F.linear(F.relu(F.linear(query, g(p + "net.0.weight"), g(p + "net.0.bias"))), g(p + "net.3.weight"), g(p + "net.3.bias"))
        query = ln(o + query, p + "norm")
        if iter_pred:
            o = head(query)
            outs.append(o)
            amasks = o[4]
    if not iter_pred:
        outs.append(head(query))
    labels, scores, masks, paras, _ = outs[-1]
    res = {"labels": labels, "scores": scores, "masks": masks, "parameters": paras}
    if iter_pred:
        res["aux_outputs"] = [{"labels": a[0], "scores": a[1], "masks": a[2], "parameters": a[3]} for a in outs[:-1]]
    return res


# ----------------------------------------------------------------------------- whole hot path (CPU baseline + parity)
def knn_torch(x, k, metric=0):
    """Literal multi-threaded torch restatement of knn / knn_points_normals (M4:30-90): per-cloud
    N x N matrix + topk.  Used for the CPU baseline timing (the C oracle is single-threaded)."""
    out = []
    with torch.no_grad():
        for b in range(x.shape[0]):
            if metric == 0:
                xb = x[b:b + 1]
                inner = -2 * torch.matmul(xb.transpose(2, 1), xb)
                xx = torch.sum(xb ** 2, dim=1, keepdim=True)
                pd = -xx - inner - xx.transpose(2, 1)
            else:
                p, n = x[b:b + 1, 0:3], x[b:b + 1, 3:6]
                xx = torch.sum(p ** 2, dim=1, keepdim=True)
                ppd = xx - 2 * torch.matmul(p.transpose(2, 1), p) + xx.transpose(2, 1)
                npd = 2 - 2 * torch.matmul(n.transpose(2, 1), n)
                pd = -(ppd * (1 + npd))
            out.append(pd.topk(k=k, dim=-1)[1])
    return torch.cat(out, 0)


def hot_path(sd, points, normals, k, knn_fn=None, idxs=None, topk_idx=None, info=None):
    """forward_train of PrimitivesEmbeddingDGCNGn up to pt_offsets (M4:634-747), functional, fp32.
    sd: state dict with the reference's parameter names.  knn_fn(x, k, metric) -> idx."""
    g = lambda n: sd[n]
    if knn_fn is None:
        knn_fn = lambda x, kk, metric: torch.from_numpy(knn_model(x.detach().numpy(), kk, kk, metric))
    B, N, _ = points.shape
    pts = torch.cat([points, normals], -1).permute(0, 2, 1)
    used = []

    def ec(inp, i, metric):
        idx = idxs[i] if idxs is not None else knn_fn(inp, k, metric)
        used.append(idx)
        w = g("encoder.conv%d.0.weight" % (i + 1))[:, :, 0, 0]
        return edgeconv_block(inp, idx, w, g("encoder.bn%d.weight" % (i + 1)), g("encoder.bn%d.bias" % (i + 1)), 2)

    x1 = ec(pts, 0, 1)
    x2 = ec(x1, 1, 0)
    x3 = ec(x2, 2, 0)
    xf = torch.cat((x1, x2, x3), 1)
    h = F.relu(F.group_norm(F.conv1d(xf, g("encoder.mlp1.weight"), g("encoder.mlp1.bias")), 8,
                            g("encoder.bnmlp1.weight"), g("encoder.bnmlp1.bias")))
    x = torch.cat([h.max(dim=2)[0].view(B, 1024, 1).repeat(1, 1, N), xf], 1)
    cgr = lambda t, c, b, G: F.relu(F.group_norm(F.conv1d(t, g(c + ".weight"), g(c + ".bias")), G, g(b + ".weight"), g(b + ".bias")))
    x = cgr(x, "conv1", "bn1", 8)
    x_all = cgr(x, "conv2", "bn2", 4)
    x_type = cgr(x_all, "mlp_prim_prob1", "bn_prim_prob1", 4)
    type_pp = F.conv1d(x_type, g("mlp_prim_prob2.weight"), g("mlp_prim_prob2.bias"))
    type_forgroup = type_pp.permute(0, 2, 1)
    type_per_point = F.log_softmax(type_pp, dim=1).permute(0, 2, 1)
    x_para = cgr(x_all, "mlp_param_prob1", "bn_param_prob1", 4)
    p = F.conv1d(x_para, g("mlp_param_prob2.weight"), g("mlp_param_prob2.bias")).transpose(1, 2)
    unit = lambda v: v / (torch.norm(v, dim=-1, keepdim=True).repeat(1, 1, 3) + 1e-12)
    param = torch.cat([p[:, :, :4], unit(p[:, :, 4:7]), p[:, :, 7:8], unit(p[:, :, 8:11]), p[:, :, 11:15],
                       unit(p[:, :, 15:18]), p[:, :, 18:22]], 2)
    nf = get_graph_feature_with_normals_g(pts, idx=used[0])       # M4:691 recomputes the identical kNN
    nf = grouped_block(nf, g("conv_normal.0.weight")[:, :, 0, 0], g("bn_normal.weight"), g("bn_normal.bias"), 2)
    x = cgr(torch.cat([x_all, x_type, x_para, nf], 1), "mlp_seg_prob1", "bn_seg_prob1", 4)
    output_feats = F.conv1d(x, g("mlp_seg_prob2.weight"), g("mlp_seg_prob2.bias")).permute(0, 2, 1)
    fp = cgr(torch.cat([x_all, pts], 1), "conv3", "bn3", 4).permute(0, 2, 1)
    off_sd = {kk[len("offset_pred_block."):]: v for kk, v in sd.items() if kk.startswith("offset_pred_block.")}
    off = offset_pred_module(pts[:, 0:3].permute(0, 2, 1), fp, output_feats, off_sd, topk_idx=topk_idx, info=info)
    if info is not None:
        info.update(x1=x1, x2=x2, x3=x3, x_all=x_all)
    return dict(type_per_point=type_per_point, param_per_point=param,
                semantic_scores=type_forgroup.reshape(-1, type_forgroup.shape[-1]),
                pt_offsets=off.permute(0, 2, 1).reshape(-1, 3), output_feats=output_feats), used


# ------------------------------------------------------------------------------------------
# Sparse convolutions of the instance tiny U-Net by their DEFINITION: a dense conv3d on the densified grid, read back at
# the active sites (softgroup/model/blocks.py:44-143 uses the un-vendored spconv package for these; "parity unpinned").
# Weights (K, Cin, Cout), K = 27 offsets (dx,dy,dz) x-major resp. 8 corners -- the layout of gcanet_amd/sparseconv.py.
# ------------------------------------------------------------------------------------------
def _densify(feats, idx, D, batch):
    out = feats.new_zeros(batch, D, D, D, feats.shape[1])
    i = idx.long()
    out[i[:, 0], i[:, 1], i[:, 2], i[:, 3]] = feats
    return out.permute(0, 4, 1, 2, 3)


def _read(dense, idx):
    i = idx.long()
    return dense.permute(0, 2, 3, 4, 1)[i[:, 0], i[:, 1], i[:, 2], i[:, 3]]


def subm_conv3(feats, idx, D, batch, W):
    """SubMConv3d(k=3, pad=1, no bias): out[x] = sum_d W[d] . in[x+d] at the active sites only."""
    Cin, Cout = W.shape[1], W.shape[2]
    w = W.view(3, 3, 3, Cin, Cout).permute(4, 3, 0, 1, 2)
    return _read(F.conv3d(_densify(feats, idx, D, batch), w, padding=1), idx)


def coarse_sites(idx):
    """Occupied 2x2x2 cells in (sample, x, y, z) order -> (M2,4)."""
    c = idx.long().clone()
    c[:, 1:] //= 2
    return torch.unique(c, dim=0).to(idx.dtype)


def strided_conv2(feats, idx, D, batch, W):
    """SparseConv3d(k=2, s=2, no bias) -> (feats2, idx2)."""
    Cin, Cout = W.shape[1], W.shape[2]
    w = W.view(2, 2, 2, Cin, Cout).permute(4, 3, 0, 1, 2)
    Dp = D + (D % 2)
    dense = F.pad(_densify(feats, idx, D, batch), (0, Dp - D, 0, Dp - D, 0, Dp - D))
    idx2 = coarse_sites(idx)
    return _read(F.conv3d(dense, w, stride=2), idx2), idx2


def inverse_conv2(feats2, idx2, D, batch, W, idx_fine):
    """SparseInverseConv3d(k=2): out[i] = W[corner(i)] . in2[parent(i)] on the fine sites of the paired strided conv."""
    Cin, Cout = W.shape[1], W.shape[2]
    w = W.view(2, 2, 2, Cin, Cout).permute(3, 4, 0, 1, 2)
    D2 = (D + 1) // 2
    up = F.conv_transpose3d(_densify(feats2, idx2, D2, batch), w, stride=2)[:, :, :D, :D, :D]
    return _read(up, idx_fine)


def _bn(x, sd, p, eps=1e-4):
    return F.batch_norm(x, None, None, sd[p + ".weight"], sd[p + ".bias"], True, 0.1, eps)


def _res_block(sd, p, x, idx, D, batch):
    h = subm_conv3(F.relu(_bn(x, sd, p + ".conv_branch.0")), idx, D, batch, sd[p + ".conv_branch.2.weight"])
    h = subm_conv3(F.relu(_bn(h, sd, p + ".conv_branch.3")), idx, D, batch, sd[p + ".conv_branch.5.weight"])
    ib = p + ".i_branch.0.weight"
    return h + (x @ sd[ib].t() if ib in sd else x)


def tiny_unet(sd, feats, idx, D, batch, prefix="tiny_unet", reps=2):
    """UBlock([C, 2C], BatchNorm1d(eps 1e-4), 2, ResidualBlock) in training mode (blocks.py:83-143)."""
    x = feats
    for i in range(reps):
        x = _res_block(sd, "%s.blocks.block%d" % (prefix, i), x, idx, D, batch)
    if (prefix + ".conv.2.weight") in sd:
        h, idx2 = strided_conv2(F.relu(_bn(x, sd, prefix + ".conv.0")), idx, D, batch, sd[prefix + ".conv.2.weight"])
        h = tiny_unet(sd, h, idx2, (D + 1) // 2, batch, prefix + ".u", reps)
        h = inverse_conv2(F.relu(_bn(h, sd, prefix + ".deconv.0")), idx2, D, batch, sd[prefix + ".deconv.2.weight"], idx)
        x = torch.cat((x, h), dim=1)
        for i in range(reps):
            x = _res_block(sd, "%s.blocks_tail.block%d" % (prefix, i), x, idx, D, batch)
    return x


# ------------------------------------------------------------------------------------------
# Loss-side callers (utils/loss_utils.py), restated literally for the parity tests of gcanet_amd/losses.py.
# ------------------------------------------------------------------------------------------
def embedding_loss(pred_feat, gt_label, t_pull=0.5, t_push=1.5):
    """loss_utils.py:203-257, one cloud and one label at a time."""
    B = pred_feat.shape[0]
    pull_total, push_total = pred_feat.new_zeros(1), pred_feat.new_zeros(1)
    for b in range(B):
        groups = [pred_feat[b][gt_label[b] == v] for v in range(-1, int(gt_label[b].max()) + 1)]
        groups = [g for g in groups if g.shape[0] > 0]
        means = [g.mean(0, keepdim=True) for g in groups]
        pull_b = sum(F.relu(torch.norm(g - m, 2, dim=1) - t_pull).mean() for g, m in zip(groups, means))
        pull_total = pull_total + pull_b / len(groups)
        if len(means) > 1:
            c = torch.cat(means, 0)
            d = torch.norm(c[:, None, :] - c[None, :, :], 2, dim=2)
            off = d[~torch.eye(c.shape[0], dtype=torch.bool)]
            push_total = push_total + F.relu(t_push - off).mean()
    pull_total, push_total = pull_total / B, push_total / B
    return pull_total + push_total, pull_total, push_total


def instance_loss(cls_scores, mask_scores, iou_scores, proposals_idx, proposals_offset, instance_labels,
                  instance_pointnum, instance_cls, instance_batch_idxs, instance_classes=10):
    """loss_utils.py:308-435 on the CPU oracle ops."""
    from . import get_mask_iou, get_mask_label
    bg, thr = instance_classes - 1, 0.5
    pidx, poff = proposals_idx[:, 1].int().numpy(), proposals_offset.int().numpy()
    il, ipn, ic = instance_labels.numpy(), instance_pointnum.numpy(), instance_cls.numpy()
    iou_c = torch.from_numpy(get_mask_iou(pidx, poff, il, ipn))
    fg = instance_cls != 0
    fg_cls = instance_cls[fg]
    n = iou_c.shape[0]
    assigned = torch.full((n,), -1, dtype=torch.long)
    mx, am = iou_c[:, fg].max(1)
    assigned[mx >= thr] = am[mx >= thr]
    labels = torch.full((n,), bg, dtype=torch.long)
    labels[assigned >= 0] = fg_cls[assigned[assigned >= 0]]
    cls_loss = F.cross_entropy(cls_scores, labels)
    mcl = labels[instance_batch_idxs.long()]
    sig = mask_scores.sigmoid()[torch.arange(mcl.shape[0]), mcl]
    ml = torch.from_numpy(get_mask_label(pidx, poff, il, ic, ipn, iou_c.numpy(), thr))
    w = (ml != -1).float()
    ml = ml.clone()
    ml[ml == -1.] = 0.5
    mask_loss = F.binary_cross_entropy(sig, ml, weight=w, reduction='sum') / (w.sum() + 1)
    ious = torch.from_numpy(get_mask_iou(pidx, poff, il, ipn, sig.detach().numpy()))
    gt_ious, _ = ious[:, fg].max(1)
    iw = (labels < bg).float()
    sl = iou_scores[torch.arange(n), labels]
    iou_loss = (F.mse_loss(sl, gt_ious, reduction='none') * iw).sum() / (iw.sum() + 1)
    return cls_loss + mask_loss + iou_loss


# ------------------------------------------------------------------------------------------
# Offset module of the reference's variant M2 (models/dgcnn-hais-concat-direct-2.py:296-462): native kNN among the key
# points + grouping_operation gathers + sigmoid KPAM.  Pinned by tests/golden/m2_offset_golden.npz (the reference's own
# source text run on the oracle's KNN / gather, tests/golden/make_golden_m2.py).
# ------------------------------------------------------------------------------------------
def offset_pred_module_m2(points, feature, semantic_feature, instance_feature, sd, nn_nb=60, n_keys=120, prefix=""):
    """points (B,N,3), feature (B,N,128), semantic_feature (B,N,Cs), instance_feature (B,N,Ci) -> offsets (B,3,N)."""
    from . import KNN_forward
    g = lambda n: torch.as_tensor(sd[prefix + n])
    B, N, _ = points.shape
    sub = key_point_indices(N, n_keys)
    key_pts, key_feat = points[:, sub], feature[:, sub]                       # (B,120,3), (B,120,128)
    _, I = KNN_forward(key_pts.detach().permute(0, 2, 1).contiguous().numpy(),
                       points.detach().permute(0, 2, 1).contiguous().numpy(), nn_nb, False)    # (B,k,N) search_knn.py:11-14
    idx = torch.from_numpy(I).permute(0, 2, 1)                                # (B,N,k) ranks among the key points
    bi = torch.arange(B).view(B, 1, 1)
    direction = key_pts[bi, idx] - points.unsqueeze(2)                        # M2:429
    f = torch.cat([key_feat[bi, idx], direction], 3)                          # (B,N,k,131)
    ins_knn = instance_feature[bi, idx]                                       # FULL cloud indexed with key ranks (M2:415)
    dist = torch.cdist(instance_feature.unsqueeze(2), ins_knn, p=2).squeeze(2)    # (B,N,k)  M2:321
    a = F.conv1d(F.relu(F.conv1d(dist.permute(0, 2, 1), g("attention_inst.conv1.0.weight"))),
                 g("attention_inst.conv1.2.weight")).permute(0, 2, 1)
    f = torch.sigmoid(a).unsqueeze(-1) * f                                    # M2:340-347
    y = F.conv2d(f.permute(0, 3, 2, 1), g("conv1.0.weight"))                  # (B,128,k,N)
    y = F.leaky_relu(F.group_norm(y, 2, g("bn1.weight"), g("bn1.bias")), 0.2).max(dim=-2)[0]
    y = torch.cat([y, feature.permute(0, 2, 1)], dim=1)
    return F.conv1d(y, g("mlp_offset.weight"), g("mlp_offset.bias"))
