"""Inference pass (reference ``procedures/infer.py:9-50``): probabilities = sigmoid(logits), collected on the host."""
from __future__ import annotations

import numpy as np
import torch

__all__ = ["predict_proba", "binary_metrics"]


def predict_proba(model, loader, device, comparative=False):
    model.eval()
    probs, labels = [], []
    with torch.no_grad():
        for graph, seq, target, prop in loader:
            if comparative:
                graph, seq, prop = tuple(g.to(device) for g in graph), tuple(s.to(device) for s in seq), tuple(p.to(device) for p in prop)
                final = model.forward_comparative(graph, seq, prop)[4]
            else:
                final = model(graph.to(device), seq.to(device), prop.to(device))[3]
            probs.append(torch.sigmoid(final).reshape(-1).cpu())
            labels.append(target.reshape(-1).cpu())
    return torch.cat(probs).numpy(), torch.cat(labels).numpy()


def binary_metrics(y_true, y_prob, threshold=0.5):
    """ROC-AUC / PR-AUC / accuracy / F1 (sklearn, as ``procedures/metric.py:43-115``) + Youden threshold."""
    from sklearn import metrics
    out = {}
    if len(np.unique(y_true)) == 2:
        fpr, tpr, thr = metrics.roc_curve(y_true, y_prob)
        out["roc_auc"] = float(metrics.auc(fpr, tpr))
        out["pr_auc"] = float(metrics.average_precision_score(y_true, y_prob))
        out["optimal_threshold"] = float(thr[int(np.argmax(tpr - fpr))])
    pred = (y_prob >= threshold).astype(np.float32)
    out["accuracy"] = float((pred == y_true).mean())
    out["f1"] = float(metrics.f1_score(y_true, pred, zero_division=0))
    return out
