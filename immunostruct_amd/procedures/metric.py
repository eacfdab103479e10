"""Binary-classification metrics of the inference path (reference ``procedures/metric.py:6-115``).

Same names, arguments, printed report and dictionary keys as the reference: ``mean_PPVn`` (mean positive predictive
value over the top-n predictions, n = 1 .. #positives, optionally only the first ``topk`` of them),
``find_optimal_threshold`` (Youden's J = TPR - FPR maximised over the ROC operating points) and ``evaluate_metrics``.
The reference takes the curves from scikit-learn; here they are a few lines of numpy on the sorted scores
(``_binary_curve``), pinned to the reference's outputs by ``tests/golden/metrics.npz``
(``oracle/make_golden_metrics.py``) and compared against scikit-learn in ``tests/test_metrics.py``.
"""
from __future__ import annotations

import numpy as np

__all__ = ["mean_PPVn", "find_optimal_threshold", "evaluate_metrics"]


def mean_PPVn(values_true, values_pred, topk=None):
    assert len(values_true) == len(values_pred)
    order = np.argsort(values_pred)[::-1]           # descending score; ties in numpy's default sort order, as the reference
    hits = np.asarray(values_true)[order]
    ppv = np.cumsum(hits) / np.arange(1, len(hits) + 1)
    positives = int(hits.sum())
    head = ppv[:positives]
    if topk is not None:
        if topk >= len(head):
            print(f"`mean_PPVn`: topk ({topk}) bigger than number of positive samples ({positives}).")
        else:
            head = head[:topk]
    return np.mean(head)


def _binary_curve(y_true, y_score):
    """cumulative (false positives, true positives, threshold) at every distinct score, scores descending"""
    y_true = np.asarray(y_true).astype(np.float64).ravel()
    y_score = np.asarray(y_score).astype(np.float64).ravel()
    if y_true.shape != y_score.shape:
        raise ValueError("targets and scores differ in length")
    order = np.argsort(y_score, kind="mergesort")[::-1]
    y_true, y_score = y_true[order], y_score[order]
    last = np.r_[np.nonzero(np.diff(y_score))[0], y_true.size - 1]      # last index of every run of equal scores
    tps = np.cumsum(y_true)[last]
    fps = 1.0 + last - tps
    return fps, tps, y_score[last]


def _roc(y_true, y_score, drop_intermediate=True):
    fps, tps, thr = _binary_curve(y_true, y_score)
    if drop_intermediate and fps.size > 2:
        # points in the interior of a straight segment change neither the area nor the corners
        keep = np.r_[True, np.logical_or(np.diff(fps, 2), np.diff(tps, 2)), True]
        fps, tps, thr = fps[keep], tps[keep], thr[keep]
    fps, tps, thr = np.r_[0.0, fps], np.r_[0.0, tps], np.r_[np.inf, thr]
    if fps[-1] <= 0 or tps[-1] <= 0:
        raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
    return fps / fps[-1], tps / tps[-1], thr


def _trapezoid(x, y):
    return float(np.sum(0.5 * (y[1:] + y[:-1]) * np.diff(x)))


def _pr_auc(y_true, y_score):
    fps, tps, _ = _binary_curve(y_true, y_score)
    precision = np.divide(tps, tps + fps, out=np.zeros_like(tps), where=(tps + fps) != 0)
    recall = tps / tps[-1] if tps[-1] > 0 else np.ones_like(tps)
    precision, recall = np.r_[precision[::-1], 1.0], np.r_[recall[::-1], 0.0]
    return -_trapezoid(recall, precision)       # recall decreases along the curve


def find_optimal_threshold(y_true, y_prob):
    fpr, tpr, thresholds = _roc(y_true, y_prob)
    optimal_threshold = thresholds[int(np.argmax(tpr - fpr))]
    assert optimal_threshold >= 0 and optimal_threshold <= 1
    return optimal_threshold


def _confusion(y_true, y_pred):
    y_true, y_pred = np.asarray(y_true).astype(bool), np.asarray(y_pred).astype(bool)
    return (int(np.sum(y_true & y_pred)), int(np.sum(~y_true & y_pred)), int(np.sum(y_true & ~y_pred)), int(np.sum(~y_true & ~y_pred)))


def _ratio(a, b):
    return float(a) / float(b) if b else 0.0


def evaluate_metrics(true_targets, predicted_probs, optimal_threshold):
    true_targets, predicted_probs = np.asarray(true_targets), np.asarray(predicted_probs)
    fpr, tpr, _ = _roc(true_targets, predicted_probs)
    out = {"optimal_threshold": optimal_threshold}
    for suffix, thr in (("", 0.5), ("_op", optimal_threshold)):
        pred = predicted_probs >= thr
        tp, fp, fn, tn = _confusion(true_targets, pred)
        out["accuracy" + suffix] = _ratio(tp + tn, tp + fp + fn + tn)
        out["f1" + suffix] = _ratio(2 * tp, 2 * tp + fp + fn)
        out["precision" + suffix] = _ratio(tp, tp + fp)
        out["recall" + suffix] = _ratio(tp, tp + fn)
        out["ppvn" + suffix] = mean_PPVn(true_targets, pred)
        out["ppv30" + suffix] = mean_PPVn(true_targets, pred, topk=30)
    out["roc_auc"] = _trapezoid(fpr, tpr)
    out["pr_auc"] = _pr_auc(true_targets, predicted_probs)

    order = ["optimal_threshold", "accuracy", "accuracy_op", "f1", "f1_op", "precision", "precision_op", "recall", "recall_op",
             "roc_auc", "pr_auc", "ppvn", "ppvn_op", "ppv30", "ppv30_op"]
    out = {k: out[k] for k in order}          # the reference's key order

    print("metrics")
    print(f"ROC AUC: {out['roc_auc']:.4f}")
    print(f"PR AUC: {out['pr_auc']:.4f}")
    for label, key in (("Accuracy", "accuracy"), ("F1 Score", "f1"), ("Precision", "precision"), ("Recall", "recall"),
                       ("Mean PPVn", "ppvn"), ("PPVn (n=30)", "ppv30")):
        print(f"{label} @0.5: {out[key]:.4f}")
        print(f"{label} @op: {out[key + '_op']:.4f}")
    return out
