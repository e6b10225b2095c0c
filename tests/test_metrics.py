"""mIoU bookkeeping against a plain numpy restatement of mmseg/core/evaluation/metrics.py:28-86."""
import numpy as np
import torch

from cmda_amd import metrics


def _np_iou(pred, label, nc, ignore):
    mask = label != ignore
    p, g = pred[mask], label[mask]
    inter = np.histogram(p[p == g], bins=nc, range=(0, nc - 1))[0] if False else np.bincount(p[p == g], minlength=nc)[:nc]
    ap, al = np.bincount(p, minlength=nc)[:nc], np.bincount(g, minlength=nc)[:nc]
    return inter.astype(np.float64), (ap + al - inter).astype(np.float64), ap.astype(np.float64), al.astype(np.float64)


def test_intersect_and_union_and_miou():
    rng = np.random.RandomState(0)
    nc = 19
    preds = [rng.randint(0, nc, (44, 64)) for _ in range(3)]
    gts = [rng.randint(0, nc, (44, 64)) for _ in range(3)]
    for g in gts:
        g[rng.rand(*g.shape) < 0.1] = 255
        g[g == 7] = 3          # class 7 absent from the ground truth
    for p in preds:
        p[p == 7] = 2          # ... and from the predictions: NaN IoU, excluded from the mean
    tot = [np.zeros(nc)] * 4
    for p, g in zip(preds, gts):
        tot = [a + b for a, b in zip(tot, _np_iou(p, g, nc, 255))]
    got = metrics.total_intersect_and_union([torch.from_numpy(p) for p in preds], [torch.from_numpy(g) for g in gts], nc, 255)
    for a, b in zip(got, tot):
        assert np.array_equal(a.numpy(), b)
    r = metrics.mean_iou([torch.from_numpy(p) for p in preds], [torch.from_numpy(g) for g in gts], nc, 255)
    iou = tot[0] / tot[1]
    assert np.isnan(r['IoU'][7].item()) and np.isnan(iou[7])
    assert abs(r['mIoU'].item() - np.nanmean(iou)) < 1e-12
    assert abs(r['aAcc'].item() - tot[0].sum() / tot[3].sum()) < 1e-12
    assert abs(r['mAcc'].item() - np.nanmean(tot[0] / tot[3])) < 1e-12
    r0 = metrics.eval_metrics([torch.from_numpy(preds[0])], [torch.from_numpy(gts[0])], nc, 255, ('mIoU', 'mDice'), nan_to_num=0)
    assert r0['IoU'][7].item() == 0 and r0['Dice'].shape == (nc,)


def test_reduce_zero_label_and_label_map():
    pred = torch.tensor([[0, 1, 2, 2]])
    gt = torch.tensor([[1, 2, 3, 0]])          # reduce_zero_label: 0 -> ignored, others shift down by one
    i, u, p, l = metrics.intersect_and_union(pred, gt, 3, 255, reduce_zero_label=True)
    assert i.tolist() == [1, 1, 1] and l.tolist() == [1, 1, 1] and p.tolist() == [1, 1, 1]
    i, u, p, l = metrics.intersect_and_union(pred, torch.tensor([[9, 1, 2, 2]]), 3, 255, label_map={9: 0})
    assert i.tolist() == [1, 1, 2]


def _gold():
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    return {k: v for k, v in np.load(os.path.join(here, 'golden', 'metrics.npz')).items()}


def _check_against_reference(device):
    """tests/golden/metrics.npz = outputs of the reference's own mmseg/core/evaluation/metrics.py (make_golden.py::metrics)"""
    g = _gold()
    nc = 19
    preds = [torch.from_numpy(g[f'pred{i}'].astype(np.int64)).to(device) for i in range(3)]
    gts = [torch.from_numpy(g[f'gt{i}'].astype(np.int64)).to(device) for i in range(3)]
    names = ('inter', 'union', 'area_pred', 'area_label')
    for k, v in zip(names, metrics.total_intersect_and_union(preds, gts, nc, 255)):
        assert np.array_equal(v.cpu().numpy(), g['tot_' + k].astype(np.float64)), k
    for k, v in zip(names, metrics.intersect_and_union(preds[0], gts[0], nc, 255)):
        assert np.array_equal(v.cpu().numpy(), g['one_' + k].astype(np.float64)), k
    r = metrics.eval_metrics(preds, gts, nc, 255, ('mIoU', 'mDice'))
    for k in ('aAcc', 'IoU', 'Acc', 'Dice'):
        np.testing.assert_allclose(r[k].cpu().numpy(), g['m_' + k], rtol=1e-6, equal_nan=True, err_msg=k)   # the reference is fp32
    assert np.isnan(g['m_IoU'][7]) and np.isnan(r['IoU'][7].item())
    r0 = metrics.eval_metrics(preds, gts, nc, 255, ('mIoU',), nan_to_num=0)
    for k in ('aAcc', 'IoU', 'Acc'):
        np.testing.assert_allclose(r0[k].cpu().numpy(), g['m0_' + k], rtol=1e-6, err_msg=k)
    for k, v in zip(names, metrics.intersect_and_union(preds[1], gts[1], nc - 1, 255, reduce_zero_label=True)):
        assert np.array_equal(v.cpu().numpy(), g['rz_' + k].astype(np.float64)), 'reduce_zero_label ' + k
    for k, v in zip(names, metrics.intersect_and_union(preds[2], gts[2], nc, 255, label_map={5: 4, 9: 255})):
        assert np.array_equal(v.cpu().numpy(), g['lm_' + k].astype(np.float64)), 'label_map ' + k
    miou = metrics.mean_iou(preds, gts, nc, 255)
    assert abs(miou['mIoU'].item() - np.nanmean(g['m_IoU'].astype(np.float64))) < 1e-6


def test_metrics_match_reference_golden_cpu():
    _check_against_reference(torch.device('cpu'))


import pytest  # noqa: E402


@pytest.mark.gpu
def test_metrics_match_reference_golden_gpu():
    if not torch.cuda.is_available():
        pytest.skip('no GPU on this machine')
    _check_against_reference(torch.device('cuda:0'))
