"""Evaluation metrics of the segmentation path ("next" row): per-class intersection / union
histograms and mIoU / mAcc / aAcc, computed on the device the predictions live on.

Reference behaviour: mmseg/core/evaluation/metrics.py:28-86 (`intersect_and_union`: pixels whose label is `ignore_index`
are dropped, then four `num_classes`-bin histograms), :89-125 (`total_intersect_and_union`: sums over images),
:128-330 (`mean_iou` / `eval_metrics`: aAcc = sum(intersect) / sum(label area), IoU = intersect / union,
Acc = intersect / label area; NaN for classes absent from both prediction and ground truth, `nan_to_num` optional).
Written from that description with `torch.bincount` instead of `torch.histc` (identical counts for integer labels)."""
import torch


def intersect_and_union(pred_label, label, num_classes, ignore_index=255, label_map=None, reduce_zero_label=False):
    """pred_label, label: integer tensors of the same shape -> (area_intersect, area_union, area_pred, area_label), each a
    float64 tensor [num_classes] on the input's device."""
    pred_label = torch.as_tensor(pred_label)
    label = torch.as_tensor(label).to(pred_label.device).long().clone()
    pred_label = pred_label.long()
    if label_map:
        src = label.clone()
        for old_id, new_id in label_map.items():
            label[src == old_id] = new_id
    if reduce_zero_label:
        label[label == 0] = 255
        label = label - 1
        label[label == 254] = 255
    mask = label != ignore_index
    pred, lab = pred_label[mask], label[mask]
    inter = pred[pred == lab]

    def hist(x):
        x = x[(x >= 0) & (x < num_classes)]
        return torch.bincount(x, minlength=num_classes).to(torch.float64)
    area_intersect, area_pred, area_label = hist(inter), hist(pred), hist(lab)
    return area_intersect, area_pred + area_label - area_intersect, area_pred, area_label


def total_intersect_and_union(results, gt_seg_maps, num_classes, ignore_index=255, label_map=None, reduce_zero_label=False):
    tot = None
    for r, g in zip(results, gt_seg_maps):
        parts = intersect_and_union(r, g, num_classes, ignore_index, label_map, reduce_zero_label)
        tot = parts if tot is None else tuple(a + b.to(a.device) for a, b in zip(tot, parts))
    return tot


def eval_metrics(results, gt_seg_maps, num_classes, ignore_index=255, metrics=('mIoU',), nan_to_num=None, label_map=None,
                 reduce_zero_label=False):
    """-> dict(aAcc=float tensor, IoU=[C], Acc=[C]) (+ Dice with 'mDice'); absent classes are NaN unless nan_to_num."""
    inter, union, pred, lab = total_intersect_and_union(results, gt_seg_maps, num_classes, ignore_index, label_map,
                                                        reduce_zero_label)
    out = {'aAcc': inter.sum() / lab.sum()}
    for m in metrics:
        if m == 'mIoU':
            out['IoU'], out['Acc'] = inter / union, inter / lab
        elif m == 'mDice':
            out['Dice'], out['Acc'] = 2 * inter / (pred + lab), inter / lab
        else:
            raise KeyError(f'metrics {m} is not supported')
    if nan_to_num is not None:
        out = {k: torch.nan_to_num(v, nan=float(nan_to_num)) for k, v in out.items()}
    return out


def mean_iou(results, gt_seg_maps, num_classes, ignore_index=255, nan_to_num=None, label_map=None, reduce_zero_label=False):
    r = eval_metrics(results, gt_seg_maps, num_classes, ignore_index, ('mIoU',), nan_to_num, label_map, reduce_zero_label)
    return {'aAcc': r['aAcc'], 'mIoU': torch.nanmean(r['IoU']), 'mAcc': torch.nanmean(r['Acc']), 'IoU': r['IoU'], 'Acc': r['Acc']}
