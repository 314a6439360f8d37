"""Evaluation metrics of the frame-level fine-tuning engine.

``calculate_metrics`` mirrors engine_for_frame_finetuning.calculate_metrics (:593-636): same inputs ([n,2] logits, [n] labels),
same return tuple.  Everything thresholded (accuracy / precision / recall / F1 / confusion matrix at 0.5, the 101-threshold MCC /
precision / recall / accuracy / F1 curves of anaysis/metrics.calculate_MORE_metrics :127-208, and the *binned* AUROC / AP / ROC / PR
of torchmetrics with ``thresholds=THRESHOLDS``) derives from ONE pass of a HIP kernel over the predictions
(``tad_threshold_histogram``: exact integer counts); the exact (sort-based) AUROC / AP that the published tables use
(anaysis/metrics.py:52-54, scikit-learn definitions) run as a device sort + prefix sums.  No CPU path for the counting.

torchmetrics is neither vendored nor version-pinned by the reference (INSTALL.md:27); the binned curves restate its published
algorithm (``_binary_precision_recall_curve_update/_compute``, ``_binary_roc_compute``, ``_auc_compute_without_check``):
parity for those is pinned only against this package's own oracle restatement, not against torchmetrics itself.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from ._lib import check

THRESHOLDS = np.arange(0.00, 1.001, 0.01).tolist()  # anaysis/metrics.py:16


def threshold_confusion(probs: torch.Tensor, labels: torch.Tensor, thresholds=THRESHOLDS) -> np.ndarray:
    """int64 [T,2,2]: confmat[t][label][pred] with pred = (p >= thresholds[t]) compared in float32 (as numpy / torch compare a
    float32 array with Python-float thresholds)."""
    if not probs.is_cuda:
        raise _lib.TadError("threshold_confusion: predictions must be on the GPU (the counting runs as a HIP kernel; no CPU path)")
    p = probs.detach().reshape(-1).float().contiguous()
    y = labels.detach().reshape(-1).to(device=p.device, dtype=torch.int32).contiguous()
    if p.numel() != y.numel() or p.numel() == 0:
        raise _lib.TadError("threshold_confusion: predictions and labels must be non-empty and of equal length")
    thr = torch.tensor(thresholds, dtype=torch.float32, device=p.device)
    if thr.numel() > 1 and not bool((thr[1:] >= thr[:-1]).all()):
        raise _lib.TadError("threshold_confusion: thresholds must be ascending")
    T = thr.numel()
    hist = torch.empty((2, T + 1), dtype=torch.int64, device=p.device)
    check(_lib.load().tad_threshold_histogram(p.data_ptr(), y.data_ptr(), thr.data_ptr(), T, p.numel(), hist.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream), "tad_threshold_histogram")
    h = hist.cpu().numpy()
    total = h.sum(axis=1, keepdims=True)                       # samples per label
    below = np.cumsum(h, axis=1)[:, :T]                        # k <= i  -> predicted negative at threshold i
    cm = np.zeros((T, 2, 2), dtype=np.int64)
    cm[:, 0, 0], cm[:, 0, 1] = below[0], total[0] - below[0]   # TN, FP
    cm[:, 1, 0], cm[:, 1, 1] = below[1], total[1] - below[1]   # FN, TP
    return cm


def _safe_div(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.divide(a, b, out=np.zeros_like(a), where=b != 0)


def thresholded_scores(cm: np.ndarray):
    """per-threshold (mcc, precision, recall, accuracy, f1) with scikit-learn's definitions and zero_division=0
    (anaysis/metrics.py:190-204)"""
    tn, fp, fn, tp = (cm[:, 0, 0].astype(np.float64), cm[:, 0, 1].astype(np.float64), cm[:, 1, 0].astype(np.float64),
                      cm[:, 1, 1].astype(np.float64))
    mcc = _safe_div(tp * tn - fp * fn, np.sqrt((tp + fp) * (tp + fn) * (tn + fp) * (tn + fn)))
    return mcc, _safe_div(tp, tp + fp), _safe_div(tp, tp + fn), _safe_div(tp + tn, tp + tn + fp + fn), _safe_div(2 * tp, 2 * tp + fp + fn)


def trapezoid(x, y) -> float:
    """sklearn.metrics.auc for monotonic x"""
    x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
    return float(np.sum((x[1:] - x[:-1]) * (y[1:] + y[:-1]) / 2.0))


def binned_curves(cm: np.ndarray, thresholds=THRESHOLDS):
    """torchmetrics' binned ROC / PR curves and their areas from the per-threshold confusion matrices (see module docstring)"""
    thr = np.asarray(thresholds, dtype=np.float32)
    tn, fp, fn, tp = cm[:, 0, 0], cm[:, 0, 1], cm[:, 1, 0], cm[:, 1, 1]
    tpr, fpr = _safe_div(tp, tp + fn)[::-1], _safe_div(fp, fp + tn)[::-1]
    auroc = trapezoid(fpr, tpr)
    precision = np.concatenate([_safe_div(tp, tp + fp), [1.0]])
    recall = np.concatenate([_safe_div(tp, tp + fn), [0.0]])
    ap = float(-np.sum((recall[1:] - recall[:-1]) * precision[:-1]))
    return auroc, ap, (precision, recall, thr), (fpr, tpr, thr[::-1].copy())


def exact_auroc_ap(probs: torch.Tensor, labels: torch.Tensor):
    """scikit-learn's roc_auc_score / average_precision_score (distinct-score thresholds, ties grouped) on the device"""
    p = probs.detach().reshape(-1).float()
    y = labels.detach().reshape(-1).to(p.device).double()
    order = torch.argsort(p, descending=True, stable=True)
    ps, ys = p[order], y[order]
    last = torch.ones_like(ps, dtype=torch.bool)
    last[:-1] = ps[1:] != ps[:-1]                                # last element of every run of equal scores
    tps = torch.cumsum(ys, 0)[last]
    fps = torch.cumsum(1.0 - ys, 0)[last]
    n_pos, n_neg = tps[-1], fps[-1]
    if float(n_pos) == 0 or float(n_neg) == 0:
        raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
    z = torch.zeros(1, dtype=tps.dtype, device=tps.device)
    tpr, fpr = torch.cat([z, tps]) / n_pos, torch.cat([z, fps]) / n_neg
    auroc = torch.sum((fpr[1:] - fpr[:-1]) * (tpr[1:] + tpr[:-1]) / 2.0)
    precision, recall = tps / (tps + fps), tps / n_pos
    ap = torch.sum((recall - torch.cat([z, recall[:-1]])) * precision)
    return float(auroc), float(ap)


def calculate_more_metrics(probs: torch.Tensor, labels: torch.Tensor):
    """anaysis/metrics.calculate_MORE_metrics (:127-208) without the two sklearn curve objects: returns
    (acc, precision, recall, f1, ap, auroc, confmat, mcc[T], precision[T], recall[T], acc[T], f1[T]).
    NOTE the 4th value: the reference's threshold loop re-uses the name ``f1_val`` (:199 overwrites :173), so what it returns as
    "F1 at 0.5" is the F1 at the LAST threshold (1.0).  Reproduced as is; the real F1 at 0.5 is ``f1[T][50]``."""
    cm = threshold_confusion(probs, labels)
    mcc, pr, rc, acc, f1 = thresholded_scores(cm)
    i5 = THRESHOLDS.index(0.5)
    auroc, ap = exact_auroc_ap(probs, labels)
    return (acc[i5], pr[i5], rc[i5], f1[-1], ap, auroc, cm[i5].tolist(), mcc.tolist(), pr.tolist(), rc.tolist(), acc.tolist(), f1.tolist())


def calculate_metrics(preds: torch.Tensor, labels: torch.Tensor, do_softmax: bool = True):
    """engine_for_frame_finetuning.calculate_metrics (:593-636): returns
    (acc, recall, precision, f1, confmat, auroc, ap, pr_curve, roc_curve, (mcc_auc, mcc_max, mcc_max_threshold, mcc_05));
    accuracy / recall / precision / F1 / confusion matrix use the arg-max class (``torch.max(preds, 1)``), the curves the class-1
    probability binned at THRESHOLDS."""
    if do_softmax:
        preds = torch.nn.functional.softmax(preds, dim=1)
    values = preds[:, 1].contiguous()
    hard = torch.max(preds, 1)[1]
    y = labels.to(preds.device)
    tp = int(((hard == 1) & (y == 1)).sum()); fp = int(((hard == 1) & (y == 0)).sum())
    fn = int(((hard == 0) & (y == 1)).sum()); tn = int(((hard == 0) & (y == 0)).sum())
    acc = float(_safe_div(tp + tn, tp + tn + fp + fn)); recall = float(_safe_div(tp, tp + fn)); precision = float(_safe_div(tp, tp + fp))
    f1 = float(_safe_div(2 * tp, 2 * tp + fp + fn))
    confmat = [[tn, fp], [fn, tp]]
    cm = threshold_confusion(values, y)
    auroc, ap, pr_curve, roc_curve = binned_curves(cm)
    mcc = thresholded_scores(cm)[0].tolist()
    mcc_max = max(mcc)
    return (acc, recall, precision, f1, confmat, auroc, ap, pr_curve, roc_curve,
            (trapezoid(THRESHOLDS, mcc), mcc_max, THRESHOLDS[mcc.index(mcc_max)], mcc[THRESHOLDS.index(0.5)]))
