"""Loss-side callers of the hot-path ops (utils/loss_utils.py of the reference), device
resident and without per-class Python loops:

  instance_loss           loss_utils.py:308-435 on gcanet_amd.softgroup.ops.get_mask_iou_on_cluster / _on_pred /
                          get_mask_label (csrc/softgroup.hip) -- same assignment rule, same three terms
  compute_embedding_loss  loss_utils.py:203-257 (pull towards the instance mean, push the means apart) as segmented
                          reductions over (cloud, label) instead of `for i in batch: for j in classes:` boolean masks
"""
import torch
import torch.nn.functional as F

from .softgroup.ops import get_mask_iou_on_cluster, get_mask_iou_on_pred, get_mask_label


def compute_embedding_loss(pred_feat, gt_label, t_pull=0.5, t_push=1.5, num_labels=None):
    """pred_feat (B,N,K) float, gt_label (B,N) int (>= -1).  Returns (loss, pull, push), each of shape (1,) as in the
    reference.  Per cloud: pull = mean over present labels of mean_i relu(|f_i - c_label| - t_pull); push = mean over
    ordered pairs of distinct present labels of relu(t_push - |c_a - c_b|), skipped when only one label is present."""
    B, N, K = pred_feat.shape
    dev = pred_feat.device
    lab = gt_label.long() + 1                                   # -1 becomes segment 0 (loss_utils.py:216-219)
    # segments per cloud; num_labels (an upper bound of gt_label.max() + 1 the data loader knows) avoids reading the
    # maximum back: a host synchronisation in the middle of every training step
    L = int(lab.max()) + 1 if num_labels is None else int(num_labels) + 1
    seg = (torch.arange(B, device=dev).view(B, 1) * L + lab).reshape(-1)        # (B*N) segment id
    f = pred_feat.reshape(B * N, K)
    cnt = torch.zeros(B * L, device=dev, dtype=f.dtype).index_add_(0, seg, torch.ones_like(seg, dtype=f.dtype))
    present = cnt > 0
    centers = torch.zeros(B * L, K, device=dev, dtype=f.dtype).index_add_(0, seg, f) / cnt.clamp(min=1).unsqueeze(1)
    dis = F.relu(torch.norm(f - centers.index_select(0, seg), 2, dim=1) - t_pull)   # index_select: backward = one index_add
    seg_mean = torch.zeros(B * L, device=dev, dtype=f.dtype).index_add_(0, seg, dis) / cnt.clamp(min=1)
    npres = present.view(B, L).sum(1).to(f.dtype)                               # labels present per cloud (>= 1)
    pull = ((seg_mean * present).view(B, L).sum(1) / npres).sum() / B
    c = centers.view(B, L, K)
    diff = c[:, :, None, :] - c[:, None, :, :]
    pair = (present.view(B, L, 1) & present.view(B, 1, L)) & ~torch.eye(L, dtype=torch.bool, device=dev)
    dst = torch.sqrt((diff * diff).sum(-1) + (~pair).to(f.dtype))              # +1 off the mask: no sqrt(0) gradient
    push_c = (F.relu(t_push - dst) * pair).sum((1, 2)) / (npres * (npres - 1)).clamp(min=1)
    push = (push_c * (npres > 1)).sum() / B
    pull, push = pull.view(1), push.view(1)
    return pull + push, pull, push


def instance_loss(cls_scores, mask_scores, iou_scores, proposals_idx, proposals_offset, instance_labels,
                  instance_pointnum, instance_cls, instance_batch_idxs, instance_classes=10):
    """loss_utils.py:308-435.  proposals_idx (S,2) / proposals_offset as forward_grouping returns them (CPU or device)."""
    ignore_label, pos_iou_thr = 0, 0.5
    bg = instance_classes - 1
    if proposals_idx.size(0) == 0 or int((instance_cls != ignore_label).sum()) == 0:
        return cls_scores.sum() * 0 + mask_scores.sum() * 0 + iou_scores.sum() * 0
    dev = cls_scores.device
    pidx = proposals_idx[:, 1].int().to(dev).contiguous()
    poff = proposals_offset.int().to(dev).contiguous()
    ious_on_cluster = get_mask_iou_on_cluster(pidx, poff, instance_labels, instance_pointnum)
    fg = instance_cls != ignore_label
    fg_cls = instance_cls[fg]
    max_iou, argmax_iou = ious_on_cluster[:, fg].max(1)
    labels = torch.where(max_iou >= pos_iou_thr, fg_cls[argmax_iou], torch.full_like(argmax_iou, bg))
    cls_loss = F.cross_entropy(cls_scores, labels)
    mask_cls_label = labels[instance_batch_idxs.long()]
    sig = mask_scores.sigmoid().gather(1, mask_cls_label.view(-1, 1)).squeeze(1)
    mask_label = get_mask_label(pidx, poff, instance_labels, instance_cls, instance_pointnum, ious_on_cluster, pos_iou_thr)
    weight = (mask_label != -1).float()
    mask_label = torch.where(mask_label == -1., torch.full_like(mask_label, 0.5), mask_label)
    mask_loss = F.binary_cross_entropy(sig, mask_label, weight=weight, reduction='sum') / (weight.sum() + 1)
    ious = get_mask_iou_on_pred(pidx, poff, instance_labels, instance_pointnum, sig.detach().contiguous())
    gt_ious, _ = ious[:, fg].max(1)
    w = (labels < bg).float()
    iou_slice = iou_scores.gather(1, labels.view(-1, 1)).squeeze(1)
    iou_score_loss = (F.mse_loss(iou_slice, gt_ious, reduction='none') * w).sum() / (w.sum() + 1)
    return cls_loss + mask_loss + iou_score_loss


class SumMeanSquaresFunction(torch.autograd.Function):
    """sum_t mean(v_t^2) over up to 8 CUDA tensors (f32 / bf16) with ONE launch each way (csrc/heads.hip:
    gcn_multi_mean_square_fwd/_bwd) -- the synthetic objective bench.py puts on the hot path's outputs.  The tensor list
    travels in the kernel arguments: nothing is uploaded, so the call can be captured into a HIP graph."""
    _done = {}

    @staticmethod
    def _host_lists(vs, grads=None):
        import ctypes as C
        n = len(vs)
        return ((C.c_void_p * n)(*[v.data_ptr() for v in vs]),
                None if grads is None else (C.c_void_p * n)(*[g.data_ptr() for g in grads]),
                (C.c_int64 * n)(*[v.numel() for v in vs]),
                (C.c_int * n)(*[int(v.dtype == torch.bfloat16) for v in vs]))

    @staticmethod
    def forward(ctx, *vs):
        from . import _lib
        assert 1 <= len(vs) <= 8
        for v in vs:
            _lib.require_cuda(v)
            assert v.dtype in (torch.float32, torch.bfloat16) and v.numel() > 0
        vs = tuple(v.contiguous() for v in vs)
        dev = vs[0].device
        pv, _, pn, pb = SumMeanSquaresFunction._host_lists(vs)
        key = (dev, _lib.stream_of(vs[0]))     # calls on one stream are ordered; another stream gets its own counter
        done = SumMeanSquaresFunction._done.get(key)
        if done is None:                       # the kernel leaves the counter zero: one 4-byte allocation per stream, for good
            done = SumMeanSquaresFunction._done[key] = torch.zeros(1, dtype=torch.int32, device=dev)
        part = torch.empty(_lib.lib().gcn_multi_mean_square_ws_chunks(pn, len(vs)), dtype=torch.float64, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        with _lib.on_device(loss):
            _lib.call("gcn_multi_mean_square_fwd", pv, pn, pb, len(vs), _lib.ptr(part), _lib.ptr(done), _lib.ptr(loss),
                      _lib.stream_of(loss))
        ctx.save_for_backward(*vs)
        return loss

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        vs = ctx.saved_tensors
        g = g.float().contiguous()
        grads = tuple(torch.empty_like(v) for v in vs)
        pv, pg, pn, pb = SumMeanSquaresFunction._host_lists(vs, grads)
        with _lib.on_device(g):
            _lib.call("gcn_multi_mean_square_bwd", pv, pg, pn, pb, len(vs), _lib.ptr(g), _lib.stream_of(g))
        return grads


def sum_mean_squares(*vs):
    """sum_t mean(v_t^2) == sum(v.float().pow(2).mean() for v in vs)."""
    return SumMeanSquaresFunction.apply(*vs)
