"""Inference passes with the reference's signatures (``procedures/infer.py:9-103``, ``procedures/infer_SSL.py:9-103``).

``inference`` / ``inference_comparative`` (and the ``*_SSL`` forms, whose loaders yield a fifth field and whose models
return a fifth output) run the model in eval mode over a loader, turn the logits into probabilities and return the
dictionary of ``procedures.metric.evaluate_metrics`` at 0.5 and at the Youden-optimal threshold (computed on this
very set unless ``optimal_threshold`` is given), plus the raw arrays on request.  Difference that does not change
results: probabilities and targets stay on the device until the loader is exhausted (one D2H copy instead of one
synchronising ``.cpu()`` per batch).  The Kaplan-Meier clinical validation (``procedures/clinical_validation.py``, needs
the clinical tables and ``lifelines``) is outside the hot-path scope: passing ``clinical_loader`` raises.
"""
from __future__ import annotations

import numpy as np
import torch

from .metric import evaluate_metrics, find_optimal_threshold

__all__ = ["inference", "inference_comparative", "inference_SSL", "inference_comparative_SSL", "predict_proba", "binary_metrics"]


def _to(device, obj):
    if isinstance(obj, (tuple, list)):
        return type(obj)(_to(device, o) for o in obj)
    return obj.to(device)


def _collect(model, data_loader, device, comparative, ssl):
    model.eval()
    probs, targets = [], []
    with torch.no_grad():
        for batch in data_loader:
            graph, seq, target, prop = _to(device, tuple(batch[:4]))       # SSL loaders carry the masked residue as a fifth field
            if comparative:
                final = model.forward_comparative(graph, seq, prop)[4]
            else:
                final = model(graph, seq, prop)[3]
            probs.append(torch.sigmoid(final).reshape(-1))
            targets.append(target.reshape(-1))
    if not probs:
        return np.zeros(0), np.zeros(0)
    # float64 like the reference's ``probs.tolist()`` -> ``np.array`` round trip
    return torch.cat(probs).cpu().numpy().astype(np.float64), torch.cat(targets).cpu().numpy()


def _run(config, model, data_loader, device, clinical_loader, return_raw_preds, fig_save_folder, optimal_threshold, comparative, ssl):
    if clinical_loader:
        raise NotImplementedError("clinical validation (Kaplan-Meier p-values) is outside this package's scope")
    predicted_probs, true_targets = _collect(model, data_loader, device, comparative, ssl)
    if optimal_threshold is None:
        optimal_threshold = find_optimal_threshold(true_targets, predicted_probs)
    output_dict = evaluate_metrics(true_targets, predicted_probs, optimal_threshold)
    if return_raw_preds:
        output_dict["predicted_probs"] = predicted_probs
        output_dict["true_targets"] = true_targets
    return output_dict


def inference(config, model, data_loader, device, clinical_loader=None, return_raw_preds=False, fig_save_folder=None,
              optimal_threshold=None):
    return _run(config, model, data_loader, device, clinical_loader, return_raw_preds, fig_save_folder, optimal_threshold, False, False)


def inference_comparative(config, model, data_loader, device, clinical_loader=None, return_raw_preds=False, fig_save_folder=None,
                          optimal_threshold=None):
    return _run(config, model, data_loader, device, clinical_loader, return_raw_preds, fig_save_folder, optimal_threshold, True, False)


def inference_SSL(config, model, data_loader, device, clinical_loader=None, return_raw_preds=False, fig_save_folder=None,
                  optimal_threshold=None):
    return _run(config, model, data_loader, device, clinical_loader, return_raw_preds, fig_save_folder, optimal_threshold, False, True)


def inference_comparative_SSL(config, model, data_loader, device, clinical_loader=None, return_raw_preds=False,
                              fig_save_folder=None, optimal_threshold=None):
    return _run(config, model, data_loader, device, clinical_loader, return_raw_preds, fig_save_folder, optimal_threshold, True, True)


# ---- short forms used by the synthetic entry points --------------------------------------------------
def predict_proba(model, loader, device, comparative=False):
    return _collect(model, loader, device, comparative, False)


def binary_metrics(y_true, y_prob, threshold=0.5):
    """ROC-AUC / PR-AUC / accuracy / F1 at ``threshold`` and the Youden threshold; a one-class set gets the counts only"""
    from . import metric as M
    out = {}
    if len(np.unique(y_true)) == 2:
        fpr, tpr, thr = M._roc(y_true, y_prob)
        out["roc_auc"] = M._trapezoid(fpr, tpr)
        out["pr_auc"] = M._pr_auc(y_true, y_prob)
        out["optimal_threshold"] = float(thr[int(np.argmax(tpr - fpr))])
    tp, fp, fn, tn = M._confusion(y_true, y_prob >= threshold)
    out["accuracy"] = M._ratio(tp + tn, tp + fp + fn + tn)
    out["f1"] = M._ratio(2 * tp, 2 * tp + fp + fn)
    return out
