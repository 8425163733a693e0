"""Drop-in for ``models/query_decoder.py`` (Mask3D-style query decoder): same classes, constructor
arguments, parameter names and output dict.  nn.MultiheadAttention parameters are kept (so reference
checkpoints load) but the attention core runs through the fused HIP kernel; the per-cloud Python loop of
the reference's cross attention (query_decoder.py:32-43) is kept because clouds have different lengths."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .attention import sdpa


def _mha(attn, q_in, k_in, v_in, attn_mask=None, precision="f32"):
    """nn.MultiheadAttention(batch_first=True).forward with the fused core.  The fused kernel never forms the
    attention probabilities, so dropout ON them (nn.MultiheadAttention(dropout=p) in training mode) cannot be
    applied: that configuration is refused rather than silently run without dropout (the reference's default is
    dropout=0.0, models/query_decoder.py:7,49)."""
    if attn.dropout > 0.0 and attn.training:
        raise RuntimeError("gcanet_amd.query_decoder: attention dropout p=%g in training mode is not supported by the "
                           "fused attention kernel; use dropout=0.0 (the reference default) or eval()" % attn.dropout)
    d, h = attn.embed_dim, attn.num_heads
    w, b = attn.in_proj_weight, attn.in_proj_bias
    q = F.linear(q_in, w[:d], b[:d])
    k = F.linear(k_in, w[d:2 * d], b[d:2 * d])
    v = F.linear(v_in, w[2 * d:], b[2 * d:])
    B, L, _ = q.shape
    S = k.shape[1]
    split = lambda t, n: t.view(B, n, h, d // h).transpose(1, 2).reshape(B * h, n, d // h)
    o = sdpa(split(q, L), split(k, S), split(v, S), attn_mask, (d // h) ** -0.5, precision)
    o = o.view(B, h, L, d // h).transpose(1, 2).reshape(B, L, d)
    return F.linear(o, attn.out_proj.weight, attn.out_proj.bias)


class CrossAttentionLayer(nn.Module):
    def __init__(self, d_model=256, nhead=8, dropout=0.0, precision="f32"):
        super().__init__()
        self.precision = precision
        self.attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout, batch_first=True)
        self.norm = nn.LayerNorm(d_model)
        self.dropout = nn.Dropout(dropout)
        self._reset_parameters()

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def with_pos_embed(self, tensor, pos):
        return tensor if pos is None else tensor + pos

    def forward(self, source, query, batch_offsets, attn_masks=None, pe=None):
        B = len(batch_offsets) - 1
        outputs = []
        query = self.with_pos_embed(query, pe)
        # Clouds of EQUAL length (a fixed-N batch, BASELINE configs[4]) go through the attention core together: the same
        # per-(cloud, head) arithmetic, but one launch of B*heads workgroups instead of B launches of `heads` workgroups
        # each streaming 16384 keys (0.2 ms apiece at 8 workgroups on 256 CUs).  Ragged batches keep the reference's loop.
        offs = [int(o) for o in batch_offsets] if not torch.is_tensor(batch_offsets) or not batch_offsets.is_cuda else None
        if offs is not None and B > 1 and len({offs[i + 1] - offs[i] for i in range(B)}) == 1 and offs[1] > offs[0]:
            kv = source[offs[0]:offs[-1]].reshape(B, offs[1] - offs[0], source.shape[-1])
            am = None
            if attn_masks:
                h = self.attn.num_heads                     # per-cloud (nq, n) masks -> one (B*heads, nq, n) mask
                am = torch.stack(list(attn_masks)).unsqueeze(1).expand(-1, h, -1, -1).reshape(B * h, *attn_masks[0].shape)
            return _mha(self.attn, query, kv, kv, am, self.precision) + query
        for i in range(B):
            kv = source[batch_offsets[i]:batch_offsets[i + 1]].unsqueeze(0)
            am = attn_masks[i] if attn_masks else None
            output = _mha(self.attn, query[i].unsqueeze(0), kv, kv, am, self.precision)
            # quirk kept: the results of self.dropout(output) and self.norm(output) are discarded (:40-42)
            outputs.append(output + query[i])
        return torch.cat(outputs, dim=0)


class SelfAttentionLayer(nn.Module):
    def __init__(self, d_model=256, nhead=8, dropout=0.0, precision="f32"):
        super().__init__()
        self.precision = precision
        self.attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout, batch_first=True)
        self.norm = nn.LayerNorm(d_model)
        self.dropout = nn.Dropout(dropout)

    def with_pos_embed(self, tensor, pos):
        return tensor if pos is None else tensor + pos

    def forward(self, x, pe=None):
        q = k = self.with_pos_embed(x, pe)
        output = _mha(self.attn, q, k, x, None, self.precision)
        return self.norm(self.dropout(output) + x)


class FFN(nn.Module):
    def __init__(self, d_model, hidden_dim, dropout=0.0, activation_fn='relu'):
        super().__init__()
        act = nn.ReLU() if activation_fn == 'relu' else nn.GELU()
        self.net = nn.Sequential(nn.Linear(d_model, hidden_dim), act, nn.Dropout(dropout),
                                 nn.Linear(hidden_dim, d_model), nn.Dropout(dropout))
        self.norm = nn.LayerNorm(d_model)

    def forward(self, x):
        return self.norm(self.net(x) + x)


class QueryDecoder(nn.Module):
    """query_decoder.py:104-239."""

    def __init__(self, num_layer=6, num_query=100, num_class=18, in_channel=32, d_model=256, nhead=8,
                 hidden_dim=1024, dropout=0.0, activation_fn='relu', iter_pred=False, attn_mask=False, pe=False,
                 precision="f32"):
        super().__init__()
        self.num_layer, self.num_query = num_layer, num_query
        self.input_proj = nn.Sequential(nn.Linear(in_channel, d_model), nn.LayerNorm(d_model), nn.ReLU())
        self.query = nn.Embedding(num_query, d_model)
        if pe:
            self.pe = nn.Embedding(num_query, d_model)
        self.cross_attn_layers = nn.ModuleList([CrossAttentionLayer(d_model, nhead, dropout, precision) for _ in range(num_layer)])
        self.self_attn_layers = nn.ModuleList([SelfAttentionLayer(d_model, nhead, dropout, precision) for _ in range(num_layer)])
        self.ffn_layers = nn.ModuleList([FFN(d_model, hidden_dim, dropout, activation_fn) for _ in range(num_layer)])
        self.out_norm = nn.LayerNorm(d_model)
        self.out_cls = nn.Sequential(nn.Linear(d_model, d_model), nn.ReLU(), nn.Linear(d_model, num_class))
        self.out_score = nn.Sequential(nn.Linear(d_model, d_model), nn.ReLU(), nn.Linear(d_model, 1))
        self.out_paras = nn.Sequential(nn.Linear(d_model, d_model), nn.ReLU(), nn.Linear(d_model, 22))
        self.x_mask = nn.Sequential(nn.Linear(in_channel, d_model), nn.ReLU(), nn.Linear(d_model, d_model))
        self.iter_pred, self.attn_mask = iter_pred, attn_mask

    def get_mask(self, query, mask_feats, batch_offsets):
        pred_masks, attn_masks = [], []
        for i in range(len(batch_offsets) - 1):
            pred_mask = torch.einsum('nd,md->nm', query[i], mask_feats[batch_offsets[i]:batch_offsets[i + 1]])
            if self.attn_mask:
                am = (pred_mask.sigmoid() < 0.5).bool()
                am[torch.where(am.sum(-1) == am.shape[-1])] = False
                attn_masks.append(am.detach())
            pred_masks.append(pred_mask)
        return pred_masks, attn_masks

    def prediction_head(self, query, mask_feats, batch_offsets):
        query = self.out_norm(query)
        pred_masks, attn_masks = self.get_mask(query, mask_feats, batch_offsets)
        return self.out_cls(query), self.out_score(query), pred_masks, self.out_paras(query), attn_masks

    def forward_simple(self, x, batch_offsets):
        inst_feats, mask_feats = self.input_proj(x), self.x_mask(x)
        B = len(batch_offsets) - 1
        query = self.query.weight.unsqueeze(0).repeat(B, 1, 1)
        for i in range(self.num_layer):
            query = self.cross_attn_layers[i](inst_feats, query, batch_offsets)
            query = self.self_attn_layers[i](query)
            query = self.ffn_layers[i](query)
        labels, scores, masks, paras, _ = self.prediction_head(query, mask_feats, batch_offsets)
        return {'labels': labels, 'parameters': paras, 'masks': masks, 'scores': scores}

    def forward_iter_pred(self, x, batch_offsets):
        outs = []
        inst_feats, mask_feats = self.input_proj(x), self.x_mask(x)
        B = len(batch_offsets) - 1
        query = self.query.weight.unsqueeze(0).repeat(B, 1, 1)
        pe = self.pe.weight.unsqueeze(0).repeat(B, 1, 1) if getattr(self, 'pe', None) else None
        o = self.prediction_head(query, mask_feats, batch_offsets)
        outs.append(o)
        attn_masks = o[4]
        for i in range(self.num_layer):
            query = self.cross_attn_layers[i](inst_feats, query, batch_offsets, attn_masks, pe)
            query = self.self_attn_layers[i](query, pe)
            query = self.ffn_layers[i](query)
            o = self.prediction_head(query, mask_feats, batch_offsets)
            outs.append(o)
            attn_masks = o[4]
        labels, scores, masks, paras, _ = outs[-1]
        return {'labels': labels, 'masks': masks, 'scores': scores, 'parameters': paras,
                'aux_outputs': [{'labels': a[0], 'masks': a[2], 'scores': a[1], 'parameters': a[3]} for a in outs[:-1]]}

    def forward(self, x, batch_offsets):
        return self.forward_iter_pred(x, batch_offsets) if self.iter_pred else self.forward_simple(x, batch_offsets)
